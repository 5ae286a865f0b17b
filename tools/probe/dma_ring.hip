// Probe: how fast can the stage-1 DMA pattern stream by itself?  Same 16-instance tiles, 4-row chunks (16 x 400 B),
// NB-slot LDS ring per wave, LDS budget LDSB per workgroup (sets waves/CU), touching one float per lane per chunk.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__device__ __forceinline__ void glds16(const float* base, uint32_t off, uint32_t lds) {
    uint32_t keep;
    asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o], %[base]\n\ts_mov_b32 m0, %[keep]"
                 : [keep] "=&s"(keep) : [o] "v"(off), [base] "s"(base), [lds] "s"(lds) : "memory");
}
template <int NV> __device__ __forceinline__ void wait_vm() {
    __builtin_amdgcn_s_waitcnt((NV & 0xF) | (0x7 << 4) | (0xF << 8) | ((NV >> 4) << 14));
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void dma4(const float* src, int stride, float* buf, int lane) {
    uint32_t lds = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)buf;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        int x = j * 64 + lane; int inst = x / 25; int pc = x - inst * 25;
        if (x < 400) glds16(src, (uint32_t)(inst * stride + pc * 4) * 4u, lds + j * 1024);
    }
}
template <int NB, int LDSB, int SECOND>
__global__ __launch_bounds__(64) void ring(const float* M, const float* J, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[LDSB / 4];
    const int lane = threadIdx.x; const size_t t0 = (size_t)blockIdx.x * 16;
    const float* Mt = M + t0 * 625; const float* Jt = J + t0 * 325;
    constexpr int NCH = 6 + 3 + (SECOND ? 3 : 0);    // 4-row chunks only (rows 0..23 of M, 0..11 of J [+ J again])
    auto issue = [&](int m) {
        if (m >= NCH) return;
        float* dst = lds + (m % NB) * 1600;
        if (m < 6) dma4(Mt + m * 100, 625, dst, lane); else dma4(Jt + ((m - 6) % 3) * 100, 325, dst, lane);
    };
#pragma unroll
    for (int m = 0; m < NB; ++m) issue(m);
    float acc = 0.f;
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        const int younger = (NCH - 1 - m) < (NB - 1) ? (NCH - 1 - m) : (NB - 1);
        if (younger == 0) wait_vm<0>(); else if (younger == 1) wait_vm<7>(); else if (younger == 2) wait_vm<14>(); else wait_vm<21>();
        acc += lds[(m % NB) * 1600 + lane * 25];
        __builtin_amdgcn_s_waitcnt(0xF | (0x7 << 4) | (0x0 << 8) | (0x3 << 14));
        asm volatile("" ::: "memory");
        issue(m + NB);
    }
    out[blockIdx.x * 64 + lane] = acc;
}
template <int NB, int LDSB, int SECOND> void run(const char* nm, const float* M, const float* J, float* out, int tiles) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((ring<NB, LDSB, SECOND>), dim3(tiles), dim3(64), 0, 0, M, J, out);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((ring<NB, LDSB, SECOND>), dim3(tiles), dim3(64), 0, 0, M, J, out);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    double bytes = (double)tiles * 16 * (2400.0 + 1200.0 * (1 + SECOND));
    printf("%-34s %7.1f us  %6.2f TB/s (bytes DMA'd %.0f MB)\n", nm, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / 1e6);
}
int main() {
    const int tiles = 4096; size_t nM = (size_t)tiles * 16 * 625, nJ = (size_t)tiles * 16 * 325;
    float *M, *J, *out; CK(hipMalloc(&M, nM * 4 * 2)); CK(hipMalloc(&J, nJ * 4 * 2)); CK(hipMalloc(&out, tiles * 64 * 4));
    CK(hipMemset(M, 0, nM * 4 * 2)); CK(hipMemset(J, 0, nJ * 4 * 2));
    run<2, 20352, 0>("ring2 20KB (8 waves/CU) 1 pass", M, J, out, tiles);
    run<2, 20352, 1>("ring2 20KB (8 waves/CU) J twice", M, J, out, tiles);
    run<3, 26752, 1>("ring3 26KB (6 waves/CU) J twice", M, J, out, tiles);
    run<3, 20352, 1>("ring3 20KB (8 waves/CU) J twice", M, J, out, tiles);
    run<4, 26752, 1>("ring4 26KB (6 waves/CU) J twice", M, J, out, tiles);
    run<2, 12800, 1>("ring2 12.8KB (12 waves/CU) J twice", M, J, out, tiles);
    run<3, 19200, 0>("ring3 19KB (8 waves/CU) 1 pass", M, J, out, tiles);
    return 0;
}
