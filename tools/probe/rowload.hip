// Probe: what HBM rate does the LOAD PATTERN of the row16 kernel reach on its own (no arithmetic), against the same bytes
// read as one contiguous stream per wave?  4 robots per wave, lane = column, rows of M (25 x 2 loads of 8 B per lane), J
// (13 rows x 2), dq / bias, poses; 12.5 KB of LDS and <= 168 registers per wave so that the occupancy is the kernel's
// (3 waves per SIMD).  usage: rowload [robots = 65536] [steps = 8] [reps = 20]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int N = 25, K = 13, NDEV = 3;
struct Ptrs { const double *M, *J, *dq, *bias, *ee, *tgt; double* u; int B; };

// `work` fp64 FMAs per lane on four independent chains (stands in for the kernel's arithmetic; ~2 400 VALU instructions per wave)
__device__ __forceinline__ void spin(double& a0, double& a1, double& a2, double& a3, const double x, const int work) {
    for (int i = 0; i < work; i += 4) {
        a0 = __builtin_fma(a0, x, 1.0); a1 = __builtin_fma(a1, x, 1.0); a2 = __builtin_fma(a2, x, 1.0); a3 = __builtin_fma(a3, x, 1.0);
    }
}

// MODE 3: loads interleaved with a quarter of the arithmetic (rows 4 ahead), three quarters of it behind them -- the kernel's
// shape.  MODE 4: every load of the wave requested up front, then all of the arithmetic.
template <int MODE>
__global__ __launch_bounds__(64, 3) void rowwork(const Ptrs* steps, const int work) {
    __shared__ double pad[1568];
    const Ptrs p = steps[blockIdx.y];
    const int lane = threadIdx.x, q = lane >> 4, l = lane & 15;
    const int b = min(blockIdx.x * 4 + q, p.B - 1);
    const double* m0 = p.M + (size_t)b * N * N + l;
    const double* m1 = l < N - 16 ? p.M + (size_t)b * N * N + 16 + l : p.M;
    const double* j0 = p.J + (size_t)b * K * N + l;
    const double* j1 = l < N - 16 ? j0 + 16 : p.J;
    double a0 = 0.1 * lane, a1 = 0.2, a2 = 0.3, a3 = 0.4;
    const double xs = 0.999;
    double jv[2 * K], mv[2 * N];
#pragma unroll
    for (int r = 0; r < K; ++r) { jv[2 * r] = j0[r * N]; jv[2 * r + 1] = j1[r * N]; }
    double acc = p.dq[(size_t)b * N + l] + p.bias[(size_t)b * N + l] + p.ee[((size_t)b * NDEV + min(l >> 2, NDEV - 1)) * 7 + (l & 3)];
    if (MODE == 4) {
#pragma unroll
        for (int j = 0; j < N; ++j) { mv[2 * j] = m0[j * N]; mv[2 * j + 1] = m1[j * N]; }
        spin(a0, a1, a2, a3, xs, work / 5);                   // ("task error": arithmetic that needs none of the big loads)
#pragma unroll
        for (int j = 0; j < 2 * N; ++j) acc += mv[j];
        spin(a0, a1, a2, a3, xs + acc * 1e-300, work - work / 5);
    } else {
        constexpr int PF = 4;
#pragma unroll
        for (int j = 0; j < PF; ++j) { mv[2 * j] = m0[j * N]; mv[2 * j + 1] = m1[j * N]; }
        spin(a0, a1, a2, a3, xs, work / 5);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (j + PF < N) { mv[2 * (j + PF)] = m0[(j + PF) * N]; mv[2 * (j + PF) + 1] = m1[(j + PF) * N]; }
            acc += mv[2 * j] + mv[2 * j + 1];
            spin(a0, a1, a2, a3, xs + acc * 1e-300, work / 100);      // a quarter of the arithmetic rides in the row loop
        }
        spin(a0, a1, a2, a3, xs + acc * 1e-300, work - work / 5 - 25 * (work / 100));
    }
#pragma unroll
    for (int r = 0; r < 2 * K; ++r) acc += jv[r];
    acc += a0 + a1 + a2 + a3;
    pad[lane] = acc;
    if (blockIdx.x * 4 + q < p.B) { p.u[(size_t)b * N + l] = pad[lane]; if (l < N - 16) p.u[(size_t)b * N + 16 + l] = acc; }
}

template <int MODE>      // 0: the kernel's pattern, rows requested PF ahead; 1: everything requested at once; 2: contiguous stream
__global__ __launch_bounds__(64, 3) void rowload(const Ptrs* steps) {
    __shared__ double pad[1568];                 // 12.5 KB: the kernel's LDS footprint
    const Ptrs p = steps[blockIdx.y];
    const int lane = threadIdx.x, q = lane >> 4, l = lane & 15;
    const int b = min(blockIdx.x * 4 + q, p.B - 1);
    double acc = 0.0;
    if (MODE == 2) {
        // the wave's 4 robots' records as contiguous 16-byte pieces: 4 * 8536 B = 34 144 B = 2 134 pieces
        const double2* src = reinterpret_cast<const double2*>(p.M) + (size_t)blockIdx.x * 2134;
#pragma unroll 8
        for (int i = lane; i < 2134; i += 64) { const double2 v = src[i]; acc += v.x + v.y; }
    } else {
        const double* m0 = p.M + (size_t)b * N * N + l;
        const double* m1 = l < N - 16 ? p.M + (size_t)b * N * N + 16 + l : p.M;
        const double* j0 = p.J + (size_t)b * K * N + l;
        const double* j1 = l < N - 16 ? j0 + 16 : p.J;
        double e = p.ee[((size_t)b * NDEV + min(l >> 2, NDEV - 1)) * 7 + (l & 3)] + p.tgt[((size_t)b * NDEV + min(l >> 2, NDEV - 1)) * 7 + (l & 3)];
        double jv[2 * K];
#pragma unroll
        for (int r = 0; r < K; ++r) { jv[2 * r] = j0[r * N]; jv[2 * r + 1] = j1[r * N]; }
        acc += e + p.dq[(size_t)b * N + l] + p.bias[(size_t)b * N + l];
        if (MODE == 1) {
            double mv[2 * N];
#pragma unroll
            for (int j = 0; j < N; ++j) { mv[2 * j] = m0[j * N]; mv[2 * j + 1] = m1[j * N]; }
#pragma unroll
            for (int j = 0; j < 2 * N; ++j) acc += mv[j];
        } else {
            constexpr int PF = 4;
            double mv[2 * N];
#pragma unroll
            for (int j = 0; j < PF; ++j) { mv[2 * j] = m0[j * N]; mv[2 * j + 1] = m1[j * N]; }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (j + PF < N) { mv[2 * (j + PF)] = m0[(j + PF) * N]; mv[2 * (j + PF) + 1] = m1[(j + PF) * N]; }
                acc += mv[2 * j] + mv[2 * j + 1];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int r = 0; r < 2 * K; ++r) acc += jv[r];
    }
    pad[lane] = acc;
    if (blockIdx.x * 4 + q < p.B) { p.u[(size_t)b * N + l] = pad[lane]; if (l < N - 16) p.u[(size_t)b * N + 16 + l] = acc; }
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 65536, S = argc > 2 ? atoi(argv[2]) : 8, reps = argc > 3 ? atoi(argv[3]) : 20;
    const int slots = 4;
    std::vector<Ptrs> h(S);
    std::vector<double*> M(slots), J(slots), dq(slots), bias(slots), ee(slots), tgt(slots), u(S);
    for (int s = 0; s < slots; ++s) {
        CHK(hipMalloc(&M[s], (size_t)B * 8536 + 4096)); CHK(hipMemset(M[s], 0, (size_t)B * 8536 + 4096));     // MODE 2 streams the whole record from here
        CHK(hipMalloc(&J[s], (size_t)B * K * N * 8)); CHK(hipMemset(J[s], 0, (size_t)B * K * N * 8));
        CHK(hipMalloc(&dq[s], (size_t)B * N * 8)); CHK(hipMemset(dq[s], 0, (size_t)B * N * 8));
        CHK(hipMalloc(&bias[s], (size_t)B * N * 8)); CHK(hipMemset(bias[s], 0, (size_t)B * N * 8));
        CHK(hipMalloc(&ee[s], (size_t)B * NDEV * 7 * 8)); CHK(hipMemset(ee[s], 0, (size_t)B * NDEV * 7 * 8));
        CHK(hipMalloc(&tgt[s], (size_t)B * NDEV * 7 * 8)); CHK(hipMemset(tgt[s], 0, (size_t)B * NDEV * 7 * 8));
    }
    for (int s = 0; s < S; ++s) { CHK(hipMalloc(&u[s], (size_t)B * N * 8)); h[s] = Ptrs{M[s % slots], J[s % slots], dq[s % slots], bias[s % slots], ee[s % slots], tgt[s % slots], u[s], B}; }
    Ptrs* d; CHK(hipMalloc(&d, S * sizeof(Ptrs))); CHK(hipMemcpy(d, h.data(), S * sizeof(Ptrs), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const dim3 grid((B + 3) / 4, S);
    const double bytes = (double)B * S * 8536.0;
    for (int mode = 0; mode < 3; ++mode) {
        for (int w = 0; w < 3; ++w) {
            if (mode == 0) hipLaunchKernelGGL(rowload<0>, grid, dim3(64), 0, 0, d);
            else if (mode == 1) hipLaunchKernelGGL(rowload<1>, grid, dim3(64), 0, 0, d);
            else hipLaunchKernelGGL(rowload<2>, grid, dim3(64), 0, 0, d);
        }
        CHK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) {
            if (mode == 0) hipLaunchKernelGGL(rowload<0>, grid, dim3(64), 0, 0, d);
            else if (mode == 1) hipLaunchKernelGGL(rowload<1>, grid, dim3(64), 0, 0, d);
            else hipLaunchKernelGGL(rowload<2>, grid, dim3(64), 0, 0, d);
        }
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        const char* nm[3] = {"kernel's pattern, M rows 4 ahead", "kernel's pattern, all of M requested at once", "same bytes as one contiguous 16-byte stream per wave"};
        printf("%-55s %8.1f us per train of %d, %6.2f TB/s (algorithmic 8 536 B per robot)\n", nm[mode], ms / reps * 1e3, S, bytes / (ms / reps * 1e-3) / 1e12);
    }
    for (int work : {1200, 2400}) {
        for (int mode = 3; mode <= 4; ++mode) {
            for (int w = 0; w < 3; ++w) {
                if (mode == 3) hipLaunchKernelGGL(rowwork<3>, grid, dim3(64), 0, 0, d, work);
                else hipLaunchKernelGGL(rowwork<4>, grid, dim3(64), 0, 0, d, work);
            }
            CHK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) {
                if (mode == 3) hipLaunchKernelGGL(rowwork<3>, grid, dim3(64), 0, 0, d, work);
                else hipLaunchKernelGGL(rowwork<4>, grid, dim3(64), 0, 0, d, work);
            }
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            printf("%d FMAs per lane, %-44s %8.1f us per train of %d, %6.2f TB/s\n", work,
                   mode == 3 ? "loads in step with the arithmetic (4 ahead)" : "every load up front, then the arithmetic", ms / reps * 1e3, S,
                   bytes / (ms / reps * 1e-3) / 1e12);
        }
    }
    return 0;
}
