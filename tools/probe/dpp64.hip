// Probe for the fp64 row16 kernel design: semantics, hazards and issue cost of v_fmac_f64_dpp / v_mov_b64_dpp with
// row_newbcast on gfx950, accuracy of v_rcp_f64 / v_rsq_f64, and DFMA throughput.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/dpp64.hip -o tools/probe/dpp64 && tools/probe/dpp64
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// out[0]: acc + bcast5(x) * y ; out[1]: acc - bcast5(x) * y (neg modifier on the DPP source);
// out[2]: hazard test, DPP source written by the previous VALU instruction, no nop;
// out[3]: same with s_nop 1; out[4]: exec-masked source lane (lane 5 of each row disabled), bound_ctrl off
__global__ void sem(double* out, const double* a, const double* b) {
    const int i = threadIdx.x;
    double x = a[i], y = b[i];
    double r0 = 1.0, r1 = 1.0, r2 = 1.0, r3 = 1.0, r4 = 1.0;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(r0) : "v"(x), "v"(y));
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(r1) : "v"(x), "v"(y));
    double t = x;
    asm volatile("v_add_f64 %1, %1, 1.0\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(r2), "+v"(t) : "v"(y));
    double t2 = x;
    asm volatile("v_add_f64 %1, %1, 1.0\n\ts_nop 1\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(r3), "+v"(t2) : "v"(y));
    unsigned long long ke;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b32 exec_lo, 0xffdfffdf\n\ts_mov_b32 exec_hi, 0xffdfffdf\n\t"
                 "v_fmac_f64_dpp %0, %2, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "s_mov_b64 exec, %1" : "+v"(r4), "=&s"(ke) : "v"(x), "v"(y));
    out[i] = r0; out[64 + i] = r1; out[128 + i] = r2; out[192 + i] = r3; out[256 + i] = r4;
}

// accuracy of the hardware reciprocal / reciprocal square root and of one correction step
__global__ void acc(double* out, const double* a) {
    const int i = threadIdx.x + blockIdx.x * blockDim.x;
    const double d = a[i];
    double r = __builtin_amdgcn_rcp(d);
    double q = __builtin_amdgcn_rsq(d);
    out[4 * i + 0] = r;
    out[4 * i + 1] = q;
    double e = fma(-d, r, 1.0);
    out[4 * i + 2] = fma(r * e, 1.0 + e, r);                 // r (1 + e + e^2)
    double t = d * q;
    double e2 = fma(-t, q, 1.0);
    out[4 * i + 3] = fma(q * e2, fma(0.375, e2, 0.5), q);    // q (1 + e/2 + 3 e^2 / 8)
}

// issue cost: N instructions of one kind in a chain of 8 independent accumulators, cycles via s_memtime
template <int KIND>
__global__ void thr(unsigned long long* cyc, double* sink, int reps) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    double x = 1.0000001, y = 0.9999999;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if (KIND == 0) {
            asm volatile("v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2\n\tv_fma_f64 %3, %8, %9, %3\n\t"
                         "v_fma_f64 %4, %8, %9, %4\n\tv_fma_f64 %5, %8, %9, %5\n\tv_fma_f64 %6, %8, %9, %6\n\tv_fma_f64 %7, %8, %9, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        } else if (KIND == 1) {
            asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %4, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %6, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        } else if (KIND == 2) {   // dependent chain through the DPP source (substitution pattern), 2 interleaved chains
            asm volatile("v_fmac_f64_dpp %0, %0, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %2, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %0, %0, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %2, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        } else if (KIND == 3) {   // v_mov_b64_dpp
            asm volatile("v_mov_b64_dpp %0, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b64_dpp %2, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %3, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b64_dpp %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %5, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b64_dpp %6, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        } else if (KIND == 4) {   // v_rcp_f64
            asm volatile("v_rcp_f64 %0, %0\n\tv_rcp_f64 %1, %1\n\tv_rcp_f64 %2, %2\n\tv_rcp_f64 %3, %3\n\t"
                         "v_rcp_f64 %4, %4\n\tv_rcp_f64 %5, %5\n\tv_rcp_f64 %6, %6\n\tv_rcp_f64 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        } else if (KIND == 5) {   // single dependent DFMA chain
            asm volatile("v_fma_f64 %0, %0, %9, %8\n\tv_fma_f64 %0, %0, %9, %8\n\tv_fma_f64 %0, %0, %9, %8\n\tv_fma_f64 %0, %0, %9, %8\n\t"
                         "v_fma_f64 %0, %0, %9, %8\n\tv_fma_f64 %0, %0, %9, %8\n\tv_fma_f64 %0, %0, %9, %8\n\tv_fma_f64 %0, %0, %9, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        } else if (KIND == 6) {   // v_pk_fma_f32 for comparison
            asm volatile("v_pk_fma_f32 %0, %8, %9, %0\n\tv_pk_fma_f32 %1, %8, %9, %1\n\tv_pk_fma_f32 %2, %8, %9, %2\n\tv_pk_fma_f32 %3, %8, %9, %3\n\t"
                         "v_pk_fma_f32 %4, %8, %9, %4\n\tv_pk_fma_f32 %5, %8, %9, %5\n\tv_pk_fma_f32 %6, %8, %9, %6\n\tv_pk_fma_f32 %7, %8, %9, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
static int run_thr(const char* name, int waves_per_simd) {
    const int reps = 2000;
    const int blocks = 256 * 4 * waves_per_simd;      // 64-thread blocks: one wave each
    unsigned long long* dc; double* ds;
    CK(hipMalloc(&dc, blocks * 8)); CK(hipMalloc(&ds, (size_t)blocks * 64 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(thr<KIND>, dim3(blocks), dim3(64), 0, 0, dc, ds, reps);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(thr<KIND>, dim3(blocks), dim3(64), 0, 0, dc, ds, reps);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> c(blocks);
    CK(hipMemcpy(c.data(), dc, blocks * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (auto v : c) mean += (double)v; mean /= blocks;
    const double ninstr = (double)reps * 8;
    printf("%-28s %d waves/SIMD: %.2f cycles per instruction per wave (s_memtime), wall %.3f ms => %.2f Ginstr/s/SIMD-slot\n", name,
           waves_per_simd, mean / ninstr, ms, ninstr * blocks / (ms * 1e-3) / 1e9 / 1024.0);
    (void)hipFree(dc); (void)hipFree(ds);
    return 0;
}

int main() {
    double ha[64], hb[64], ho[320];
    for (int i = 0; i < 64; ++i) { ha[i] = 100.0 + i; hb[i] = 0.5 + 0.001 * i; }
    double *da, *db, *dout;
    CK(hipMalloc(&da, sizeof ha)); CK(hipMalloc(&db, sizeof hb)); CK(hipMalloc(&dout, sizeof ho));
    CK(hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, dout, da, db);
    CK(hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost));
    int bad[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 64; ++i) {
        const double src = ha[(i / 16) * 16 + 5];
        bad[0] += ho[i] != fma(src, hb[i], 1.0);
        bad[1] += ho[64 + i] != fma(-src, hb[i], 1.0);
        bad[2] += ho[128 + i] != fma(src + 1.0, hb[i], 1.0);
        bad[3] += ho[192 + i] != fma(src + 1.0, hb[i], 1.0);
    }
    printf("row_newbcast:5 semantics: plain %d bad, neg-src %d bad, hazard(no nop) %d bad, hazard(s_nop 1) %d bad of 64\n", bad[0], bad[1], bad[2], bad[3]);
    printf("exec-masked source lane: lane 0 -> %.6f (1.0 = not written, %.6f = source read anyway, other = 0-source), lane 5 -> %.6f\n",
           ho[256], fma(ha[5], hb[0], 1.0), ho[256 + 5]);

    const int NA = 1 << 16;
    std::vector<double> xa(NA), xo(4 * NA);
    for (int i = 0; i < NA; ++i) xa[i] = exp(-20.0 + 40.0 * (double)i / NA) * (1.0 + 0.37 * (double)(i % 97) / 97.0);
    double *dxa, *dxo;
    CK(hipMalloc(&dxa, NA * 8)); CK(hipMalloc(&dxo, 4 * NA * 8));
    CK(hipMemcpy(dxa, xa.data(), NA * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(acc, dim3(NA / 256), dim3(256), 0, 0, dxo, dxa);
    CK(hipMemcpy(xo.data(), dxo, 4 * NA * 8, hipMemcpyDeviceToHost));
    double e[4] = {0, 0, 0, 0};
    for (int i = 0; i < NA; ++i) {
        const double r = 1.0 / xa[i], q = 1.0 / sqrt(xa[i]);
        e[0] = fmax(e[0], fabs(xo[4 * i] - r) / r); e[1] = fmax(e[1], fabs(xo[4 * i + 1] - q) / q);
        e[2] = fmax(e[2], fabs(xo[4 * i + 2] - r) / r); e[3] = fmax(e[3], fabs(xo[4 * i + 3] - q) / q);
    }
    printf("max rel err: v_rcp_f64 %.3e, v_rsq_f64 %.3e, rcp + 1 step %.3e, rsq + 1 step %.3e\n", e[0], e[1], e[2], e[3]);

    for (int w : {1, 2, 4}) {
        if (run_thr<0>("v_fma_f64", w)) return 1;
        if (run_thr<1>("v_fmac_f64_dpp", w)) return 1;
        if (run_thr<2>("v_fmac_f64_dpp dep(4 apart)", w)) return 1;
        if (run_thr<3>("v_mov_b64_dpp", w)) return 1;
        if (run_thr<4>("v_rcp_f64", w)) return 1;
        if (run_thr<5>("v_fma_f64 dependent", w)) return 1;
        if (run_thr<6>("v_pk_fma_f32", w)) return 1;
    }
    return 0;
}
