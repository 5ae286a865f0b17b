// Probe: how many workgroups of a given shape does a CU of gfx950 really hold at once?  Every wave spins for a fixed wall-clock
// time and records where (HW_ID: XCC / SE / CU / SIMD) and when (s_memrealtime, 100 MHz) it ran; the host counts, per CU, the
// largest number of waves in flight at one moment, and the launch's span against the spin time.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/occ.hip -o tools/probe/occ && tools/probe/occ
// (The FROMQ OSC kernel: 256 threads, 166 registers, 49 584 B of LDS -- three blocks per CU on paper; its waves' residency times
// the number of waves over the kernel time says 2.1 per SIMD.)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Rec { unsigned long long t0, t1; unsigned hw, xcc; };

template <int NREG>
__global__ __launch_bounds__(256) void spin(Rec* out, int ticks, int jitter) {
    extern __shared__ double lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    // occupy NREG vector registers
    if constexpr (NREG > 128) asm volatile("v_mov_b32 v165, 0" ::: "v165");
    else if constexpr (NREG > 64) asm volatile("v_mov_b32 v100, 0" ::: "v100");
    lds[threadIdx.x] = (double)t0;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int my = ticks + (jitter ? (int)((wave * 2654435761u) >> 16) % jitter : 0);
    unsigned long long t1;
    do { __builtin_amdgcn_s_sleep(8); t1 = __builtin_amdgcn_s_memrealtime(); } while ((long long)(t1 - t0) < my);
    if ((threadIdx.x & 63) == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[wave] = Rec{t0, t1, hw, xcc};
    }
    if (lds[threadIdx.x ^ 1] < 0.0) out[0].t0 = 0;      // keep the LDS alive
}

template <int NREG>
int run(const char* what, int threads, int lds_bytes, int blocks_per_cu, int jitter) {
    const int ncu = 256, blocks = ncu * blocks_per_cu, waves = blocks * threads / 64, ticks = 2000;   // 20 us
    Rec* d;
    CK(hipMalloc((void**)&d, sizeof(Rec) * waves));
    CK(hipFuncSetAttribute((const void*)spin<NREG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(spin<NREG>, dim3(blocks), dim3(threads), lds_bytes, 0, d, ticks, jitter);
    CK(hipDeviceSynchronize());
    std::vector<Rec> h(waves);
    CK(hipMemcpy(h.data(), d, sizeof(Rec) * waves, hipMemcpyDeviceToHost));
    CK(hipFree(d));
    unsigned long long tmin = ~0ull, tmax = 0;
    std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev;      // per CU: (time, +1 / -1)
    double res = 0;
    for (const Rec& r : h) {
        tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1);
        res += (double)(r.t1 - r.t0);
        const unsigned cu = (r.xcc & 0xf) << 16 | (r.hw & 0x0000ff00u) | ((r.hw >> 13) & 0x7) << 4;      // xcc, cu_id + sh, se
        ev[cu].push_back({r.t0, +1}); ev[cu].push_back({r.t1, -1});
    }
    int peak = 0; double avg_peak = 0;
    for (auto& kv : ev) {
        std::sort(kv.second.begin(), kv.second.end());
        int cur = 0, pk = 0;
        for (auto& e : kv.second) { cur += e.second; pk = std::max(pk, cur); }
        peak = std::max(peak, pk); avg_peak += pk;
    }
    const double span = (double)(tmax - tmin) / 100.0;
    printf("%-44s %3d thr %6d B LDS %3d regs: %3zu CUs seen, peak waves/CU max %2d mean %.1f; span %.1f us for %d blocks/CU of %.0f us "
           "=> %.2f blocks in flight per CU on average\n", what, threads, lds_bytes, NREG, ev.size(), peak, avg_peak / ev.size(), span,
           blocks_per_cu, res / waves / 100.0, res / 100.0 / (threads / 64) / span / ncu);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1) {      // occ <threads> <lds bytes> ...: LDS allocation granularity (how many one-wave blocks of that size fit?)
        const int thr = atoi(argv[1]);
        for (int i = 2; i < argc; ++i) run<166>("argv", thr, atoi(argv[i]), 24 * 256 / thr, 0);
        return 0;
    }
    for (int j : {0, 1000}) {
        printf("-- jitter %d ticks\n", j);
        run<166>("FROMQ shape", 256, 49584, 24, j);
        run<166>("FROMQ shape, 40 KB", 256, 40000, 24, j);
        run<166>("two waves per block", 128, 25184, 48, j);
        run<166>("one wave per block", 64, 10976, 96, j);
        run<166>("dense shape", 64, 12544, 96, j);
        run<64>("few registers, 49 584 B", 256, 49584, 24, j);
        run<64>("few registers, 32 KB", 256, 32768, 24, j);
    }
    return 0;
}
