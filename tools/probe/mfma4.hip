// Probe: layout and rate of v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 outer products per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    int lane = threadIdx.x;
    float a = 1.0f + lane;          // A[i] for block: ?
    float b = 100.0f * (1 + lane);  // B[j]
    v4f c = {0, 0, 0, 0};
    v4f d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = d[i];
}
__global__ void rate(float* out, int iters) {
    int lane = threadIdx.x;
    float a = 1.0f + lane * 1e-3f, b = 1.0f - lane * 1e-3f;
    v4f c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 64 + lane] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void chain(float* out, int iters) {   // B operand taken from the previous result (substitution-like chain)
    int lane = threadIdx.x;
    float a = 1e-3f * lane;
    v4f c0 = {1, 1, 1, 1};
    for (int it = 0; it < iters; ++it) c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, c0[it & 3], c0, 0, 0, 0);
    out[blockIdx.x * 64 + lane] = c0[0];
}
int main() {
    float *o; hipMalloc(&o, 1 << 22); float h[256];
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o); hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
    // expect block q = lane/4: D[i][j] = A[i]*B[j]; which lane/reg holds what?
    for (int lane = 0; lane < 8; ++lane) printf("lane %d: %g %g %g %g\n", lane, h[lane*4], h[lane*4+1], h[lane*4+2], h[lane*4+3]);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 8;
    hipLaunchKernelGGL(rate, dim3(blocks), dim3(64), 0, 0, o, 10);
    hipEventRecord(e0); hipLaunchKernelGGL(rate, dim3(blocks), dim3(64), 0, 0, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)blocks * iters * 4;   // mfma per... waves = blocks (1 wave each), 8 waves/CU => 2 per SIMD
    printf("rate: %.3f ms, %.2f cycles/mfma/SIMD at 2.4GHz (2 waves per SIMD)\n", ms, ms * 1e-3 * 2.4e9 / (n / 1024.0));
    hipEventRecord(e0); hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("dependent chain (D -> B operand): %.1f cycles per mfma per wave\n", ms * 1e-3 * 2.4e9 / iters / 2);
    return 0;
}
