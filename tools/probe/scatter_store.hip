// Probe: what does a store cost when every lane of a wave writes into its OWN record (lane stride = one record), the
// access pattern of the lane-per-instance front end (csrc/osc_frontend_lane.hpp)?  1024 waves (one per SIMD), each lane
// writes N doubles:  mode 0: a run of N consecutive doubles per lane, 8-byte stores; mode 1: the same with 16-byte stores;
// mode 2: the same bytes, coalesced (lane l writes element e*64 + l of the wave's block);  mode 3: run of 8-byte stores with
// ~40 dependent FMAs between stores (stores spread in time, as in the kernel); mode 4: 8-byte stores hopping between 25 rows
// (stride 200 B) of the record, one element per row per round (the "mirror entry" pattern).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/scatter_store.hip -o tools/probe/scatter_store && tools/probe/scatter_store
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int REC = 1067;      // doubles per record (8 536 B)
constexpr int N = 448;

template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, double seed) {
    const int lane = threadIdx.x;
    const size_t inst = (size_t)blockIdx.x * 64 + lane;
    double v = seed + lane;
    if constexpr (MODE == 0) {
        double* p = out + inst * REC;
#pragma unroll 16
        for (int e = 0; e < N; ++e) p[e] = v + e;
    } else if constexpr (MODE == 1) {
        double2* p = reinterpret_cast<double2*>(out + inst * REC + (inst & 1));     // records are 8-byte aligned only
#pragma unroll 16
        for (int e = 0; e < N / 2; ++e) p[e] = double2{v + e, v - e};
    } else if constexpr (MODE == 2) {
        double* p = out + (size_t)blockIdx.x * 64 * REC + lane;
#pragma unroll 16
        for (int e = 0; e < N; ++e) p[(size_t)e * 64] = v + e;
    } else if constexpr (MODE == 3) {
        double* p = out + inst * REC;
        for (int e = 0; e < N; ++e) {
#pragma unroll
            for (int t = 0; t < 40; ++t) v = fma(v, 1.0000001, 1e-9);
            p[e] = v;
        }
    } else if constexpr (MODE == 7 || MODE == 8 || MODE == 9) {
        // the epilogue's pattern: one robot's 512 contiguous bytes per instruction, robots REC7 doubles apart, 7 chunks each
        constexpr int REC7 = MODE == 7 ? 625 : (MODE == 8 ? 640 : 625);      // 5000 B (8-byte aligned) / 5120 B (128-byte aligned)
        double* p = out + (size_t)blockIdx.x * 64 * REC7 + lane;
        if constexpr (MODE == 9) {                    // robot-major: all chunks of a robot, then the next robot
            for (int it = 0; it < 64; ++it)
#pragma unroll
                for (int ch = 0; ch < 7; ++ch) p[(size_t)it * REC7 + ch * 64] = v + it;
        } else {
            for (int ch = 0; ch < 7; ++ch)
#pragma unroll 8
                for (int it = 0; it < 64; ++it) p[(size_t)it * REC7 + ch * 64] = v + it;
        }
    } else if constexpr (MODE == 10) {
        // line-aligned emission of 8-byte-aligned records: robot li's windows start at 48 c - (li mod 16) doubles, 48 lanes each
        double* p = out + (size_t)blockIdx.x * 64 * 625;
        for (int ch = 0; ch < 14; ++ch)
#pragma unroll 8
            for (int it = 0; it < 64; ++it) {
                const int f = 48 * ch - (it & 15) + lane;
                if (lane < 48 && f >= 0 && f < 625) p[(size_t)it * 625 + f] = v + it;
            }
    } else if constexpr (MODE == 5) {                 // rows of 25, the first 13 of each written as a run, the rest never (gaps)
        double* p = out + inst * REC;
#pragma unroll 2
        for (int r = 0; r < 34; ++r) {
#pragma unroll
            for (int c = 0; c < 13; ++c) p[r * 25 + c] = v + c;
        }
    } else if constexpr (MODE == 6) {                 // the same rows written whole (zeros included)
        double* p = out + inst * REC;
#pragma unroll 2
        for (int r = 0; r < 34; ++r) {
#pragma unroll
            for (int c = 0; c < 25; ++c) p[r * 25 + c] = c < 13 ? v + c : 0.0;
        }
    } else {
        double* p = out + inst * REC;
        for (int c = 0; c < N / 25; ++c) {
#pragma unroll
            for (int r = 0; r < 25; ++r) {
#pragma unroll
                for (int t = 0; t < 40; ++t) v = fma(v, 1.0000001, 1e-9);
                p[r * 25 + c] = v;
            }
        }
    }
}

template <int MODE>
int run(double* d, const char* what) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int waves : {256, 1024, 2048}) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, d, 1.0);
        CK(hipEventRecord(a));
        const int reps = 10;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, d, 1.0);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        ms /= reps;
        const double bytes = (double)waves * 64 * N * 8;
        printf("%-58s waves=%5d  %8.1f us  %7.1f GB/s  %6.1f cycles/store-instr/wave@2.4GHz\n", what, waves, ms * 1e3, bytes / ms / 1e6,
               ms * 1e-3 * 2.4e9 / N);
    }
    return 0;
}

int main() {
    double* d;
    CK(hipMalloc(&d, (size_t)2048 * 64 * REC * 8 + 64));
    run<2>(d, "coalesced 8 B x 64 lanes");
    run<0>(d, "per-lane run, 8-byte stores");
    run<1>(d, "per-lane run, 16-byte stores");
    run<3>(d, "per-lane run, 8 B, 40 FMAs between stores");
    run<4>(d, "per-lane, 25 rows x 1 element per round, 40 FMAs between");
    run<7>(d, "512 B per robot per instr, robots 5000 B apart, chunk-major");
    run<8>(d, "512 B per robot per instr, robots 5120 B apart, chunk-major");
    run<9>(d, "512 B per robot per instr, robots 5000 B apart, robot-major");
    run<10>(d, "384 B line-aligned windows per robot per instr (625 dbl/robot; rate per 448)");
    run<5>(d, "per-lane, 34 rows: 13 of 25 written, 12 skipped (N=442)");
    run<6>(d, "per-lane, 34 rows written whole (850 stores; rate per 448)");
    return 0;
}
