// Hardware probe (not part of the product): LDS-DMA with 4-byte-aligned sources, quad DPP broadcast.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

// each wave copies, for 16 instances, 25 pieces of 16 B starting at src + inst*2500 + row0*100 bytes
__global__ void dma16(const float* __restrict__ src, float* __restrict__ dst, int row0) {
    __shared__ __attribute__((aligned(16))) float buf[16 * 100 + 64 * 4];
    int lane = threadIdx.x;
    size_t tile = blockIdx.x;
    const char* base = (const char*)src + tile * 16 * 2500 + (size_t)row0 * 100;
    // 400 pieces = 6.25 wave-instructions
    for (int j = 0; j < 7; ++j) {
        int x = j * 64 + lane;           // piece index, instance-major
        int inst = x / 25, pc = x % 25;
        if (inst > 15) { inst = 15; pc = 24; }
        const char* g = base + inst * 2500 + pc * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(buf + j * 256), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) etc
    __syncthreads();
    for (int e = lane; e < 1600; e += 64) dst[tile * 1600 + e] = buf[e];
}
__global__ void dma4(const float* __restrict__ src, float* __restrict__ dst, int row0) {
    __shared__ float buf[16 * 25 + 64];
    int lane = threadIdx.x;
    size_t tile = blockIdx.x;
    const float* base = src + tile * 16 * 625 + row0 * 25;
    for (int j = 0; j < 7; ++j) {
        int x = j * 64 + lane;
        int inst = x / 25, e = x % 25;
        if (inst > 15) { inst = 15; e = 24; }
        const float* g = base + inst * 625 + e;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(buf + j * 64), 4, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int e = lane; e < 400; e += 64) dst[tile * 400 + e] = buf[e];
}
template <int G> __device__ float qb(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), G * 0x55, 0xf, 0xf, false));
}
__global__ void dpp(float* out) {
    float v = (float)threadIdx.x;
    out[threadIdx.x] = qb<0>(v) + 100.f * qb<1>(v) + 10000.f * qb<3>(v);
}
int main() {
    const int tiles = 4096; // 65536 instances
    size_t nf = (size_t)tiles * 16 * 625;
    std::vector<float> h(nf);
    for (size_t i = 0; i < nf; ++i) h[i] = (float)(i % 1000003);
    float *d, *o; CK(hipMalloc(&d, nf * 4)); CK(hipMalloc(&o, (size_t)tiles * 1600 * 4));
    CK(hipMemcpy(d, h.data(), nf * 4, hipMemcpyHostToDevice));
    std::vector<float> r((size_t)tiles * 1600);
    for (int row0 : {0, 4, 20}) {
        hipLaunchKernelGGL(dma16, dim3(tiles), dim3(64), 0, 0, d, o, row0);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (int t = 0; t < tiles; ++t) for (int i = 0; i < 16; ++i) for (int e = 0; e < 100; ++e)
            if (r[(size_t)t * 1600 + i * 100 + e] != h[((size_t)t * 16 + i) * 625 + row0 * 25 + e]) ++bad;
        printf("dma16 row0=%d mismatches=%zu\n", row0, bad);
    }
    hipLaunchKernelGGL(dma4, dim3(tiles), dim3(64), 0, 0, d, o, 24);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(r.data(), o, (size_t)tiles * 400 * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int t = 0; t < tiles; ++t) for (int i = 0; i < 16; ++i) for (int e = 0; e < 25; ++e)
        if (r[(size_t)t * 400 + i * 25 + e] != h[((size_t)t * 16 + i) * 625 + 24 * 25 + e]) ++bad;
    printf("dma4 mismatches=%zu\n", bad);
    // timing: dma16 over all 6 chunk positions
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int it = 0; it < 20; ++it) for (int c = 0; c < 6; ++c) hipLaunchKernelGGL(dma16, dim3(tiles), dim3(64), 0, 0, d, o, c * 4);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("dma16: %.3f ms per (6 chunks x 65536 inst) => read %.1f GB/s (+ equal write)\n", ms / 20, 65536.0 * 2400 / (ms / 20 * 1e-3) / 1e9);
    float* po; CK(hipMalloc(&po, 256)); float hp[64];
    hipLaunchKernelGGL(dpp, dim3(1), dim3(64), 0, 0, po); CK(hipMemcpy(hp, po, 256, hipMemcpyDeviceToHost));
    printf("dpp lane5 -> %.0f (expect 4+500+70000=70504), lane 62 -> %.0f (expect 60+6100+630000=636160)\n", hp[5], hp[62]);
    return 0;
}
