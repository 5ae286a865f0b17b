// Throughput OSC kernel for the Dual-UR5 shapes (n = 25): FOUR LANES PER ROBOT INSTANCE.
//
// Why not one wavefront per instance: with n = 25 and k <= 13 a 64-lane wave is 60-80 % idle in
// every phase and every operand has to be broadcast, so the VALU issue rate — not HBM — bounds
// the generic kernel at ~5e7 steps/s.  Here a wave carries 16 instances; the 4 lanes of a quad own
// the joint rows i = 4s + g (slot s = 0..6, g = lane & 3; rows 25..27 are zero padding) and keep
// their rows of the Cholesky factor L, of Y = L^-1 J^T and of J in VGPRs with COMPILE-TIME indices
// (everything below is fully unrolled).  The only cross-lane traffic is quad-local DPP
// (quad_perm broadcasts / butterflies), which costs no LDS and no extra issue slot when fused.
//
// Data movement: inputs stay in the caller's batch-major AoS layout (include/irlosc.h).  Each
// wave streams its 16-instance tile HBM -> LDS with the asynchronous LDS-DMA path
// (global_load_lds_dwordx4, 4-row chunks of M and J = 16 x 400 B segments; single rows with
// global_load_lds_dword) through a 3-deep ring, with counted s_waitcnt vmcnt(N) so that two
// chunks (12.5 KB) are always in flight per wave while the current one is consumed.  The LDS image
// of a chunk is instance-major with a 100-dword stride: quad q reads dwords 100q + 4s + g, which
// spreads a 32-lane ds_read_b32 group over all 32 banks.
//
// Math per instance (same algebra as osc_generic.hpp; osc.py:41-200):
//   column-by-column (left-looking) Cholesky of M while its rows stream in, Mdq accumulated from
//   the same reads; per Jacobian row r: y_r = L^-1 J_r^T by column-oriented substitution, dx_r;
//   A = Y^T Y by quad-reduced partial dot products; then per lane (replicated in the quad):
//   Cholesky of A, W = L_A^-1, the cond(A) certificate, t = W^T W w; finally
//   u = u0 + bias - kvn*Mdq - J^T t from the register copy of J.
// Instances whose k x k solve is not certifiably the reference's branch (cond bound >= 1e5 with
// |det| < 1e-4, or A not positive definite) are appended to a worklist and redone by the generic
// kernel, which owns the eigen-decomposition.
#pragma once
#include "osc_common.hpp"
#include "osc_generic.hpp"

namespace irlosc {

namespace grp {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int N = 25;        // joints
constexpr int S = 7;         // row slots per lane (4 * 7 = 28 >= 25)
constexpr int TILE = 16;     // instances per wave
constexpr int BUF_FLOATS = 1792;   // 7 DMA instructions x 1 KiB
constexpr int NBUF = 3;
constexpr int NLISTS = 32;   // sharded worklists (counter l owns worklist[l * list_cap ...))

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    // mov_dpp (no `old` operand): with all rows/banks enabled every lane has a valid source, and not
    // materialising an `old` value saves a v_mov + the VALU->DPP wait state per broadcast
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// broadcast lane g of each quad to the whole quad
__device__ __forceinline__ float qbcast(float v, int g) {
    switch (g & 3) {
        case 0: return dpp_f<0x00>(v);
        case 1: return dpp_f<0x55>(v);
        case 2: return dpp_f<0xAA>(v);
        default: return dpp_f<0xFF>(v);
    }
}
// sum over the 4 lanes of a quad (result in every lane)
__device__ __forceinline__ float qsum(float v) {
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
    return v;
}
__device__ __forceinline__ int qbcast_i(int v, int g) {
    switch (g & 3) {
        case 0: return __builtin_amdgcn_mov_dpp(v, 0x00, 0xf, 0xf, true);
        case 1: return __builtin_amdgcn_mov_dpp(v, 0x55, 0xf, 0xf, true);
        case 2: return __builtin_amdgcn_mov_dpp(v, 0xAA, 0xf, 0xf, true);
        default: return __builtin_amdgcn_mov_dpp(v, 0xFF, 0xf, 0xf, true);
    }
}
__device__ __forceinline__ uint32_t qor(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);
    return v;
}

// s_waitcnt with only vmcnt constrained (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14])
template <int NV>
__device__ __forceinline__ void wait_vm() {
    static_assert(NV >= 0 && NV < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((NV & 0xF) | (0x7 << 4) | (0xF << 8) | ((NV >> 4) << 14));
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wait_lgkm0() {
    __builtin_amdgcn_s_waitcnt(0xF | (0x7 << 4) | (0x0 << 8) | (0x3 << 14));
    asm volatile("" ::: "memory");
}

// ---- LDS-DMA issue (global_load_lds_*), hand-counted -----------------------------------------------
// Issued through inline asm on purpose: hipcc treats the builtin form as a store to LDS that may
// alias every later ds_read and drains it with s_waitcnt vmcnt(0), which would serialise the ring.
// In asm the compiler neither counts nor waits for these loads; the kernel's own wait_vm<N>() calls
// are the only synchronisation (guide §5.7: M0 written in the same statement, s_nop 0 before use).
// Address form: SGPR-pair base (wave-uniform) + 32-bit VGPR byte offset.  LDS destination:
// M0 + lane * size, M0 advanced by `step` bytes per instruction.
#define IRLOSC_GLDS7(OP, STEP)                                                                         \
    asm volatile("s_mov_b32 %[keep], m0\n\t"                                                            \
                 "s_mov_b32 m0, %[lds]\n\t"                                                             \
                 "s_nop 0\n\t" OP " %[o0], %[base]\n\t"                                                \
                 "s_add_u32 m0, m0, " STEP "\n\ts_nop 0\n\t" OP " %[o1], %[base]\n\t"                  \
                 "s_add_u32 m0, m0, " STEP "\n\ts_nop 0\n\t" OP " %[o2], %[base]\n\t"                  \
                 "s_add_u32 m0, m0, " STEP "\n\ts_nop 0\n\t" OP " %[o3], %[base]\n\t"                  \
                 "s_add_u32 m0, m0, " STEP "\n\ts_nop 0\n\t" OP " %[o4], %[base]\n\t"                  \
                 "s_add_u32 m0, m0, " STEP "\n\ts_nop 0\n\t" OP " %[o5], %[base]\n\t"                  \
                 "s_add_u32 m0, m0, " STEP "\n\ts_nop 0\n\t" OP " %[o6], %[base]\n\t"                  \
                 "s_mov_b32 m0, %[keep]"                                                                \
                 : [keep] "=&s"(keep)                                                                   \
                 : [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [o3] "v"(o[3]), [o4] "v"(o[4]),      \
                   [o5] "v"(o[5]), [o6] "v"(o[6]), [base] "s"(base), [lds] "s"(lds)                     \
                 : "memory", "scc")

__device__ __forceinline__ uint32_t lds_addr(const float* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}

// DMA a 4-row chunk (100 floats = 25 x 16 B per instance, 16 instances): 7 wave instructions.
// src = first float of (tile instance 0, row0), wave-uniform; stride = floats between instances.
__device__ __forceinline__ void dma_rows4(const float* src, int stride, float* buf, int lane) {
    uint32_t o[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        int x = j * 64 + lane;               // 16-byte piece index, instance-major (25 per instance)
        x = x > 399 ? 399 : x;
        int inst = (x * 5243) >> 17;         // x / 25 for x < 2^12
        int pc = x - inst * 25;
        o[j] = (uint32_t)(inst * stride + pc * 4) * 4u;
    }
    const float* base = src;
    uint32_t lds = lds_addr(buf), keep;
    IRLOSC_GLDS7("global_load_lds_dwordx4", "0x400");
}
// DMA a single-row chunk (25 floats per instance): 7 wave instructions of 4 B per lane.
__device__ __forceinline__ void dma_rows1(const float* src, int stride, float* buf, int lane) {
    uint32_t o[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        int x = j * 64 + lane;
        x = x > 399 ? 399 : x;
        int inst = (x * 5243) >> 17;
        int e = x - inst * 25;
        o[j] = (uint32_t)(inst * stride + e) * 4u;
    }
    const float* base = src;
    uint32_t lds = lds_addr(buf), keep;
    IRLOSC_GLDS7("global_load_lds_dword", "0x100");
}
// DMA a contiguous block of `pieces` 16-byte pieces (pieces <= 128): 2 wave instructions.
__device__ __forceinline__ void dma_linear2(const float* src, int pieces, float* buf, int lane) {
    int x0 = lane, x1 = 64 + lane;
    x0 = x0 >= pieces ? pieces - 1 : x0;
    x1 = x1 >= pieces ? pieces - 1 : x1;
    const uint32_t o0 = (uint32_t)x0 * 16u, o1 = (uint32_t)x1 * 16u;
    const float* base = src;
    uint32_t lds = lds_addr(buf), keep;
    asm volatile("s_mov_b32 %[keep], m0\n\t"
                 "s_mov_b32 m0, %[lds]\n\t"
                 "s_nop 0\n\tglobal_load_lds_dwordx4 %[o0], %[base]\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o1], %[base]\n\t"
                 "s_mov_b32 m0, %[keep]"
                 : [keep] "=&s"(keep)
                 : [o0] "v"(o0), [o1] "v"(o1), [base] "s"(base), [lds] "s"(lds)
                 : "memory", "scc");
}

// Row rr of J chunk jc for quad q.  Chunks 0,1,2 (4 rows, stride 100) sit in ring slots 1,2,0;
// chunk 3 (1 row, stride 25) in the tail buffer.
__device__ __forceinline__ const float* jrow_ptr(const float* ring, const float* jtail, int jc, int q, int rr) {
    if (jc < 3) return ring + ((jc + 1) % 3) * BUF_FLOATS + q * 100 + rr * N;
    return jtail + q * N;
}

}  // namespace grp

// One wave = 16 instances.  K = stacked task rows, NDEV = target devices (<= 4, one quad lane each).
template <int K, int NDEV>
__global__ __launch_bounds__(64, 1) void osc_group_kernel_f32(const KParams<float> p, int32_t* __restrict__ worklist,
                                                             int32_t* __restrict__ workcount,
                                                             float* __restrict__ side, int side_cap, int list_cap) {
    using namespace grp;
    constexpr int NCHM = 7;                                   // M chunks: 6 x 4 rows + 1 row
    constexpr int NCHJ = (K + 3) / 4;                         // J chunks of 4 rows; the last may be shorter
    static_assert(K % 4 == 0 || K % 4 == 1, "the last J chunk must be 4 rows or 1 row");
    constexpr int VEC_DQ = 0, VEC_BIAS = 512, VEC_EE = 1024, VEC_TGT = 1536, VEC_TV = 2048, VEC_WR = 2560;
    constexpr int VEC_W = 3072;                               // [16][16] exchange area for the task vector
    constexpr int VEC_X = 3328;                               // [16][48] per-quad Mdq (25) and dx (K) parking
    __shared__ __attribute__((aligned(16))) float ring[NBUF * BUF_FLOATS];
    __shared__ __attribute__((aligned(16))) float vec[3328 + 768];
    __shared__ __attribute__((aligned(16))) float jtail[448];   // 1-row chunk J[12] (k = 13): 7 x 64 dwords
    constexpr bool HASJ3 = (K % 4) == 1;

    const int lane = threadIdx.x;
    const int g = lane & 3, q = lane >> 2;
    const int tile = blockIdx.x;
    const int b = tile * TILE + q;
    const size_t t0 = (size_t)tile * TILE;
    const bool has_tv = p.tvel != nullptr;
    const bool has_wr = (p.cfgflags & IRLOSC_ADMITTANCE) && p.wrench != nullptr;
    unsigned long long ts[8];
#define IRLOSC_TS(i) ts[i] = p.dbg ? __builtin_readcyclecounter() : 0ull
    IRLOSC_TS(0);

    // ---------------- prologue: vectors + first chunks in flight -----------------------------------
    // Issue order (7 DMA instructions per chunk):  vec(12) M0 M1 M2 | M3 | M4 | M5 | M6 [J3] | J0 | J1 | J2
    // where "| X" means X is issued right after the chunk three places earlier has been consumed.
    // M chunk c lives in ring slot c % 3 and is recycled; J chunks 0,1,2 land in slots 1,2,0 once
    // M4,M5,M6 are consumed and then STAY (J is re-read for u -= J^T t); the 1-row chunk J3 (k = 13)
    // has its own small buffer.  vmcnt retires in order, so "wait until at most n younger DMA
    // instructions are outstanding" is exact.
    dma_linear2(p.dq + t0 * N, TILE * N / 4, vec + VEC_DQ, lane);
    dma_linear2((p.cfgflags & IRLOSC_USE_G) ? p.bias + t0 * N : p.dq + t0 * N, TILE * N / 4, vec + VEC_BIAS, lane);
    dma_linear2(p.ee + t0 * NDEV * 7, TILE * NDEV * 7 / 4, vec + VEC_EE, lane);
    dma_linear2(p.tgt + t0 * NDEV * 7, TILE * NDEV * 7 / 4, vec + VEC_TGT, lane);
    dma_linear2(has_tv ? p.tvel + t0 * NDEV * 6 : p.dq + t0 * N, TILE * NDEV * 6 / 4, vec + VEC_TV, lane);
    dma_linear2(has_wr ? p.wrench + t0 * NDEV * 6 : p.dq + t0 * N, TILE * NDEV * 6 / 4, vec + VEC_WR, lane);
    const float* Mt = p.M + t0 * (N * N);
    const float* Jt = p.J + t0 * (K * N);
    dma_rows4(Mt, N * N, ring + 0 * BUF_FLOATS, lane);
    dma_rows4(Mt + 4 * N, N * N, ring + 1 * BUF_FLOATS, lane);
    dma_rows4(Mt + 8 * N, N * N, ring + 2 * BUF_FLOATS, lane);

    // ---------------- register state ----------------------------------------------------------------
    // Row slots 0..5 are kept as PAIRS (slots 2p, 2p+1 in one float2) so that the multiply-adds below
    // are v_pk_fma_f32: a lone wave per SIMD issues one VALU op per quad-cycle, so halving the
    // instruction count matters more than anything else.  Slot 6 (row 24, real only for g == 0) stays
    // scalar: pairing it with an all-padding slot 7 cost ~60 registers and pushed L into AGPRs.
    constexpr int P = 3;
    v2f Lp[P][24];         // strictly-lower rows of L owned by this lane (slots 0..5); upper/diagonal = 0
    float L6[24];          // row 24 (g == 0), zeros elsewhere
    v2f DinvP[P];          // 1 / L[i][i] for the lane's own rows
    float Dinv6 = 0.f;
    v2f mdqP[P];           // (M dq)[i] for own rows
    float mdq6 = 0.f;
    v2f dqP[P];
    float dq6;
    v2f Yp[K][P];          // own rows of Y = L^-1 J^T
    float Y6[K];
    uint32_t flags = 0;
#pragma unroll
    for (int pp = 0; pp < P; ++pp) { DinvP[pp] = v2f{0.f, 0.f}; mdqP[pp] = v2f{0.f, 0.f}; }

    const bool pad6 = g != 0;            // slot 6 is padding except on quad lane 0
    const int col6 = pad6 ? 0 : 24;      // safe in-range column for the masked slot
    float* xq = vec + VEC_X + q * 48;    // per-quad exchange: [0..24] Mdq, [25..37] dx

    // read one 25-float row (quad q): slots 0..5 into 3 pairs, slot 6 into a scalar (padding -> 0)
    auto load_row = [&](const float* row, v2f (&dst)[P], float& d6) {
#pragma unroll
        for (int pp = 0; pp < P; ++pp) dst[pp] = v2f{row[8 * pp + g], row[8 * pp + 4 + g]};
        const float t6 = row[col6];
        d6 = pad6 ? 0.f : t6;
    };

    wait_vm<21>();                        // the 12 vector DMAs have landed (3 chunks = 21 still in flight)
    IRLOSC_TS(1);
    load_row(vec + VEC_DQ + q * N, dqP, dq6);

    // ---------------- stream M: Cholesky column by column -----------------------------------------------
#pragma unroll
    for (int ch = 0; ch < NCHM; ++ch) {
        float* buf = ring + (ch % NBUF) * BUF_FLOATS;
        if (ch <= 3) wait_vm<14>(); else wait_vm<7 * (HASJ3 ? 3 : 2)>();
        const int R = ch < 6 ? 4 : 1;
        const int istride = R * N;
        // all rows of the chunk are read up front (one LDS round trip per chunk instead of per column)
        v2f mrow[4][P];
        float mrow6[4], dqj[4];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            load_row(buf + q * istride + rr * N, mrow[rr], mrow6[rr]);
            dqj[rr] = vec[VEC_DQ + q * N + ch * 4 + rr];
        }
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int j = ch * 4 + rr;
            const int sj = j >> 2, gj = j & 3, pj = sj >> 1;     // row j lives in slot sj of quad lane gj
            const v2f dq2 = v2f{dqj[rr], dqj[rr]};
#pragma unroll
            for (int pp = 0; pp < P; ++pp) mdqP[pp] = __builtin_elementwise_fma(mrow[rr][pp], dq2, mdqP[pp]);   // M symmetric
            mdq6 = fmaf(mrow6[rr], dqj[rr], mdq6);
            // left-looking column j: acc = M[j][i] - sum_{c<j} L[i][c] L[j][c] for the rows i >= j
            v2f acc[P];
            float acc6 = mrow6[rr];
#pragma unroll
            for (int pp = 0; pp < P; ++pp) acc[pp] = mrow[rr][pp];
#pragma unroll
            for (int c = 0; c < j; ++c) {
                const float own = sj == 6 ? L6[c] : ((sj & 1) ? Lp[pj < P ? pj : 0][c].y : Lp[pj < P ? pj : 0][c].x);
                const float ljs = -qbcast(own, gj);
                const v2f lj = v2f{ljs, ljs};
#pragma unroll
                for (int pp = pj; pp < P; ++pp) acc[pp] = __builtin_elementwise_fma(Lp[pp][c], lj, acc[pp]);
                acc6 = fmaf(L6[c], ljs, acc6);
                // keep the scalar slot-6 chain in step with the packed chains: left alone, the scheduler
                // sinks it to the end of the column and every broadcast value stays live (-> scratch)
                asm volatile("" : "+v"(acc6), "+v"(acc[P - 1]));
            }
            const float dsel = sj == 6 ? acc6 : ((sj & 1) ? acc[pj < P ? pj : 0].y : acc[pj < P ? pj : 0].x);
            float d = qbcast(dsel, gj);
            const bool notpd = !(d > 0.f);
            flags |= notpd ? IRLOSC_FLAG_M_NOT_PD : 0u;
            const float dfix = (d == d && d != 0.f) ? fabsf(d) : 1.f;
            d = notpd ? dfix : d;
            const float dinv = __builtin_amdgcn_rsqf(d);
            const v2f dinv2 = v2f{dinv, dinv};
            const bool own_row = (g == gj);
            const float below = (g > gj) ? 1.f : 0.f;
            if (j < 24) {
#pragma unroll
                for (int pp = pj + 1; pp < P; ++pp) Lp[pp][j] = acc[pp] * dinv2;
                if (sj < 6) {   // the pair that contains slot sj: rows above / on the diagonal get exact zeros
                    const v2f sc2 = acc[pj < P ? pj : 0] * dinv2;
                    if (sj & 1) Lp[pj < P ? pj : 0][j] = v2f{0.f, sc2.y * below};
                    else Lp[pj < P ? pj : 0][j] = v2f{sc2.x * below, sc2.y};
                }
                L6[j] = acc6 * dinv;             // row 24 > j always; padding lanes carry exact zeros
            }
            if (sj < 6) {
                if (sj & 1) DinvP[pj < P ? pj : 0].y = own_row ? dinv : DinvP[pj < P ? pj : 0].y;
                else DinvP[pj < P ? pj : 0].x = own_row ? dinv : DinvP[pj < P ? pj : 0].x;
            } else {
                Dinv6 = own_row ? dinv : 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // recycle the ring slot just consumed
        wait_lgkm0();
        if (ch < 3) dma_rows4(Mt + (ch + 3) * 4 * N, N * N, buf, lane);
        else if (ch == 3) {
            dma_rows1(Mt + 24 * N, N * N, buf, lane);
            if (HASJ3) dma_rows1(Jt + 12 * N, K * N, jtail, lane);
        } else dma_rows4(Jt + (ch - 4) * 4 * N, K * N, buf, lane);   // J0,J1,J2 -> slots 1,2,0 (resident)
        __builtin_amdgcn_sched_barrier(0);
    }
    IRLOSC_TS(2);
    // park Mdq in LDS (own rows; rows >= 25 never written)
#pragma unroll
    for (int pp = 0; pp < P; ++pp) { xq[8 * pp + g] = mdqP[pp].x; xq[8 * pp + 4 + g] = mdqP[pp].y; }
    if (!pad6) xq[24] = mdq6;

    // ---------------- J rows: dx and forward substitutions ------------------------------------------------
    // The R rows of a chunk are substituted together: R independent dependency chains per column step.
#pragma unroll
    for (int jc = 0; jc < NCHJ; ++jc) {
        if (jc == 0) wait_vm<14>(); else if (jc == 1) wait_vm<7>(); else wait_vm<0>();
        constexpr int RMAX = 4;
        const int R = jc < 3 ? 4 : 1;
        v2f bb[RMAX][P];
        float b6[RMAX];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int r = jc * 4 + rr;
            load_row(jrow_ptr(ring, jtail, jc, q, rr), bb[rr], b6[rr]);
            v2f dx2 = v2f{0.f, 0.f};
#pragma unroll
            for (int pp = 0; pp < P; ++pp) dx2 = __builtin_elementwise_fma(bb[rr][pp], dqP[pp], dx2);
            const float dxp = qsum(fmaf(b6[rr], dq6, dx2.x + dx2.y));
            if (g == 0) xq[25 + r] = dxp;
        }
        // column-oriented substitution: y_c = b_c / L[c][c] (owner lane), then b_i -= L[i][c] y_c
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
#pragma unroll
            for (int c = 0; c < 24; ++c) {
                const int sc = c >> 2, gc = c & 3, pc = sc >> 1;
                const float own = (sc & 1) ? bb[rr][pc].y * DinvP[pc].y : bb[rr][pc].x * DinvP[pc].x;
                const float ycs = -qbcast(own, gc);
                const v2f yc = v2f{ycs, ycs};
#pragma unroll
                for (int pp = pc; pp < P; ++pp) bb[rr][pp] = __builtin_elementwise_fma(Lp[pp][c], yc, bb[rr][pp]);
                b6[rr] = fmaf(L6[c], ycs, b6[rr]);
                asm volatile("" : "+v"(b6[rr]), "+v"(bb[rr][P - 1]));
            }
#pragma unroll
            for (int pp = 0; pp < P; ++pp) Yp[jc * 4 + rr][pp] = bb[rr][pp] * DinvP[pp];
            Y6[jc * 4 + rr] = b6[rr] * Dinv6;
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    IRLOSC_TS(3);
    // ---------------- task-space signal: lane g of the quad handles device g ---------------------------------------
    float kvn = 0.f;
    if (p.cfgflags & IRLOSC_NULLSPACE) kvn = p.null_kv[p.gains_per_instance ? b : 0];
    const float* gbase = p.gains + (p.gains_per_instance ? (size_t)b * NDEV * IRLOSC_GAIN_WORDS : 0);
    float* wls = vec + VEC_W + q * 16;
    int brA_own = 1;
    float kv_own = 0.f;
    __builtin_amdgcn_wave_barrier();
    if (g < NDEV) {
        const DevMeta dm = p.dev[g];
        const float* gg = gbase + g * IRLOSC_GAIN_WORDS;
        float gl[IRLOSC_GAIN_WORDS];
#pragma unroll
        for (int i = 0; i < IRLOSC_GAIN_WORDS; ++i) gl[i] = gg[i];
        kv_own = gl[1];
        float ee[7], tg[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            ee[i] = vec[VEC_EE + (q * NDEV + g) * 7 + i];
            tg[i] = vec[VEC_TGT + (q * NDEV + g) * 7 + i];
        }
        float e[6];
        task_error6<float>(ee, tg, dm.calc & 1u, dm.calc & 2u, e);
        apply_gains6<float>(gl, e);
        float tv[6];
        bool all_nonzero = has_tv;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            tv[i] = has_tv ? vec[VEC_TV + (q * NDEV + g) * 6 + i] : 0.f;
            all_nonzero = all_nonzero && (tv[i] != 0.f);
        }
        brA_own = all_nonzero ? 0 : 1;
        if (all_nonzero) {
            flags |= IRLOSC_FLAG_VEL_BRANCH_B;
            if (dm.jidx0 + dm.rows > K) flags |= IRLOSC_FLAG_BAD_JIDX;
        }
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (dm.dofmask & (1u << i)) {
                float v = e[i];
                if (all_nonzero) {
                    const int row = dm.jidx0 + cnt;
                    const float dxv = xq[25 + (row < K ? row : 0)];
                    const float damp = (i < 3) ? gl[6 + i] : 1.f;
                    v += gl[1] * ((row < K ? dxv : 0.f) - tv[i]) * damp;
                }
                if (has_wr) v += vec[VEC_WR + (q * NDEV + g) * 6 + i];
                wls[dm.row0 + cnt] = v;
                ++cnt;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    wait_lgkm0();
    float w[K];
#pragma unroll
    for (int r = 0; r < K; ++r) w[r] = wls[r] - kvn * xq[25 + r];

    IRLOSC_TS(4);
    // ---------------- A = Y^T Y (lower), replicated in the quad ---------------------------------------------------
    float A[K][K];
#pragma unroll
    for (int r = 0; r < K; ++r) {
#pragma unroll
        for (int s2 = 0; s2 <= r; ++s2) {
            v2f a2 = Yp[r][0] * Yp[s2][0];
#pragma unroll
            for (int pp = 1; pp < P; ++pp) a2 = __builtin_elementwise_fma(Yp[r][pp], Yp[s2][pp], a2);
            A[r][s2] = qsum(fmaf(Y6[r], Y6[s2], a2.x + a2.y));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // Park A in LDS (the ee/tgt/tvel/wrench regions are dead by now): lane g stores entries e = g mod 4.
    // Read back only by flagged quads, which hand A and w to the second stage.
    float* aq = vec + VEC_EE + q * 96;
    {
        int e = 0;
#pragma unroll
        for (int r = 0; r < K; ++r) {
#pragma unroll
            for (int c2 = 0; c2 <= r; ++c2) {
                if ((e & 3) == g) aq[e] = A[r][c2];
                ++e;
            }
        }
    }

    IRLOSC_TS(5);
    // ---------------- k x k (per lane): Cholesky of A in place, cond certificate, t = A^-1 w ------------------
    float nA2 = 0.f;
#pragma unroll
    for (int r = 0; r < K; ++r) {
#pragma unroll
        for (int c = 0; c < r; ++c) nA2 = fmaf(2.f * A[r][c], A[r][c], nA2);
        nA2 = fmaf(A[r][r], A[r][r], nA2);
    }
    bool pdA = true;
    float detA = 1.f;
    float dA[K];                       // 1 / L_A[j][j]
#pragma unroll
    for (int j = 0; j < K; ++j) {
        float d = A[j][j];
#pragma unroll
        for (int c = 0; c < j; ++c) d = fmaf(-A[j][c], A[j][c], d);
        const bool npd = !(d > 0.f);
        pdA = pdA && !npd;
        const float dfix = (d == d && d != 0.f) ? fabsf(d) : 1.f;
        d = npd ? dfix : d;
        detA *= d;
        const float di = __builtin_amdgcn_rsqf(d);
        dA[j] = di;
#pragma unroll
        for (int i = j + 1; i < K; ++i) {
            float a = A[i][j];
#pragma unroll
            for (int c = 0; c < j; ++c) a = fmaf(-A[i][c], A[j][c], a);
            A[i][j] = a * di;
        }
    }
    // ||L_A^-1||_F^2 = trace(A^-1): column j of W = L_A^-1 by forward substitution, used and dropped
    float nW2 = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        float wc[K];
        wc[j] = dA[j];
        nW2 = fmaf(wc[j], wc[j], nW2);
#pragma unroll
        for (int i = j + 1; i < K; ++i) {
            float s2 = 0.f;
#pragma unroll
            for (int c = j; c < i; ++c) s2 = fmaf(A[i][c], wc[c], s2);
            wc[i] = -dA[i] * s2;
            nW2 = fmaf(wc[i], wc[i], nW2);
        }
    }
    const bool small_det = !(fabsf(detA) >= 1e-4f);
    const float cond_bound = sqrtf(nA2) * nW2;        // >= cond_2(A) for SPD A
    const bool plain = pdA && t_finite(cond_bound) && (!small_det || cond_bound < 0.99e5f);
    flags |= small_det ? IRLOSC_FLAG_PINV_BRANCH : 0u;
    float t[K];
    // forward: z = L_A^-1 w ; backward: t = L_A^-T z
#pragma unroll
    for (int i = 0; i < K; ++i) {
        float s2 = w[i];
#pragma unroll
        for (int c = 0; c < i; ++c) s2 = fmaf(-A[i][c], t[c], s2);
        t[i] = s2 * dA[i];
    }
#pragma unroll
    for (int i = K - 1; i >= 0; --i) {
        float s2 = t[i];
#pragma unroll
        for (int c = i + 1; c < K; ++c) s2 = fmaf(-A[c][i], t[c], s2);
        t[i] = s2 * dA[i];
    }

    IRLOSC_TS(6);
    // Reserve worklist slots for the flagged quads of this wave: ONE atomic per wave on one of NLISTS
    // sharded counters (a single counter saturates at ~90 atomics/us), issued here so that its round trip
    // overlaps the torque phase.
    const unsigned long long fmask = __ballot(!plain && g == 0);
    int wl_base = 0;
    const int wl_list = blockIdx.x & (NLISTS - 1);
    if (fmask != 0ull && lane == 0) wl_base = atomicAdd(workcount + wl_list, __popcll(fmask));
    // ---------------- joint torques for the own rows ------------------------------------------------------------------
    const float* biasv = vec + VEC_BIAS + q * N;
    bool bad = false;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int i = 4 * s + g;
        const bool valid = (s < 6) || !pad6;
        const int icol = valid ? i : 0;
        const float mdq_i = xq[icol];
        float uu = 0.f;
#pragma unroll
        for (int d = 0; d < NDEV; ++d) {
            const int brA_d = qbcast_i(brA_own, d);
            const float kv_d = qbcast(kv_own, d);
            if (brA_d && (p.dev[d].joint_mask & (1u << i))) uu = -kv_d * mdq_i;
        }
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < K; ++r) acc = fmaf(jrow_ptr(ring, jtail, r >> 2, q, r & 3)[icol], t[r], acc);
        uu -= plain ? acc : 0.f;             // flagged instances keep u_base; stage 2 subtracts J^T t
        if (p.cfgflags & IRLOSC_USE_G) uu += biasv[icol];
        uu -= kvn * mdq_i;
        if (valid) {
            p.u[(size_t)b * N + i] = uu;
            bad = bad || !t_finite(uu);
        }
    }
    flags |= bad ? IRLOSC_FLAG_NONFINITE : 0u;
    flags = grp::qor(flags);
    if (fmask != 0ull) wl_base = __builtin_amdgcn_readfirstlane(wl_base);
    if (!plain) {
        flags |= IRLOSC_FLAG_EIGEN_PATH;
        // rank of this quad among the wave's flagged quads (bits of fmask below this quad's g==0 lane)
        const int rank = __popcll(fmask & ((1ull << (lane & ~3)) - 1ull));
        const int pos = wl_list * list_cap + wl_base + rank;
        if (g == 0) worklist[pos] = b;
        __builtin_amdgcn_wave_barrier();
        wait_lgkm0();
        // side[e][pos]: A (K(K+1)/2 lower entries, row-major) then w (K)
        constexpr int NA = K * (K + 1) / 2;
        for (int e = g; e < NA; e += 4) side[(size_t)e * side_cap + pos] = aq[e];
#pragma unroll
        for (int r = 0; r < K; ++r)
            if ((r & 3) == g) side[(size_t)(NA + r) * side_cap + pos] = w[r];
    }
    if (g == 0) p.flags[b] = flags;
    IRLOSC_TS(7);
    if (p.dbg && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) p.dbg[(size_t)blockIdx.x * 8 + i] = ts[i];
    }
#undef IRLOSC_TS
}

// ---------------------------------------------------------------------------------------------------------
// Second stage for the group kernel: truncated pseudo-inverse solve t = pinv(A, rcond 1e-5) w for the
// instances the first stage could not certify (osc.py:51-55 when |det A| < 1e-4 and cond(A) may exceed
// 1e5), then u -= J^T t.  4 lanes per instance like stage 1 (k x k math replicated in the quad, the
// J^T t update split by joint rows).
//
// Method (A = J M^-1 J^T is symmetric positive semi-definite, k <= 13):
//   * factor A + sigma I = L L^T (sigma = 0, raised to ~2e-6 ||A||_F only if a pivot fails in fp32);
//   * lambda_max by 12 power iterations through the factor (A x = L (L^T x) - sigma x);
//   * the eigenpairs below the cut 1e-5 lambda_max one at a time by inverse iteration with
//     deflation (at most 3; typically exactly one: a nearly rank-deficient Jacobian stack);
//   * t = P (A + sigma I)^-1 P w with P the projector off those eigenvectors (one refinement step
//     when sigma > 0), which equals sum over the kept eigenpairs of v v^T w / lambda.
// Instances with more than 3 sub-threshold eigenvalues are handed to the generic kernel (Jacobi).
template <int K>
__global__ __launch_bounds__(64) void osc_group_stage2_f32(const KParams<float> p, const int32_t* __restrict__ worklist,
                                                          const int32_t* __restrict__ workcount,
                                                          const float* __restrict__ side, int side_cap, int list_cap,
                                                          int32_t* __restrict__ worklist2, int32_t* __restrict__ workcount2) {
    using namespace grp;
    constexpr int NA = K * (K + 1) / 2;
    const int lane = threadIdx.x, g = lane & 3, q = lane >> 2;
    // block -> (list = blockIdx % NLISTS, tile within the list strided by gridDim / NLISTS)
    const int list = blockIdx.x & (NLISTS - 1);
    const int count = workcount[list];
    const int ntile = (count + TILE - 1) / TILE;
    for (int tile = blockIdx.x / NLISTS; tile < ntile; tile += gridDim.x / NLISTS) {
        const int lpos = tile * TILE + q;
        const bool live = lpos < count;
        const int posc = list * list_cap + (live ? lpos : count - 1);
        const int b = worklist[posc];
        float w[K], L[K][K], Ld[K], Li[K];     // factor: strictly-lower L, diagonal Ld, inverse diagonal Li
#pragma unroll
        for (int r = 0; r < K; ++r) w[r] = side[(size_t)(NA + r) * side_cap + posc];
        float sigma = 0.f, hi = 0.f, det = 1.f;
        bool ok = false;
        for (int attempt = 0; attempt < 4; ++attempt) {
            float nA2 = 0.f;
            int e = 0;
#pragma unroll
            for (int r = 0; r < K; ++r) {
#pragma unroll
                for (int c = 0; c <= r; ++c) {
                    const float a = side[(size_t)e * side_cap + posc];
                    ++e;
                    L[r][c] = a;
                    nA2 = fmaf(c < r ? 2.f * a : a, a, nA2);
                }
            }
            hi = sqrtf(nA2);
            bool okk = true;
            float dd = 1.f;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                float d = L[j][j] + sigma;
#pragma unroll
                for (int c = 0; c < j; ++c) d = fmaf(-L[j][c], L[j][c], d);
                const bool npd = !(d > 1e-7f * hi * 1e-1f);
                okk = okk && !npd;
                d = npd ? hi : d;
                dd *= d;
                const float di = __builtin_amdgcn_rsqf(d);
                Li[j] = di;
                Ld[j] = d * di;
#pragma unroll
                for (int i = j + 1; i < K; ++i) {
                    float a = L[i][j];
#pragma unroll
                    for (int c = 0; c < j; ++c) a = fmaf(-L[i][c], L[j][c], a);
                    L[i][j] = a * di;
                }
            }
            if (attempt == 0) det = okk ? dd : 0.f;
            ok = okk;
            if (!__any(!okk)) break;
            if (!okk) sigma = (sigma == 0.f) ? 2e-6f * hi : sigma * 8.f;
        }
        // y = (A + sigma I)^-1 x, in place
        auto solve = [&](float (&x)[K]) {
#pragma unroll
            for (int i = 0; i < K; ++i) {
                float s2 = x[i];
#pragma unroll
                for (int c = 0; c < i; ++c) s2 = fmaf(-L[i][c], x[c], s2);
                x[i] = s2 * Li[i];
            }
#pragma unroll
            for (int i = K - 1; i >= 0; --i) {
                float s2 = x[i];
#pragma unroll
                for (int c = i + 1; c < K; ++c) s2 = fmaf(-L[c][i], x[c], s2);
                x[i] = s2 * Li[i];
            }
        };
        // y = A x = L (L^T x) - sigma x
        auto amul = [&](const float (&x)[K], float (&y)[K]) {
            float z[K];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                float s2 = Ld[j] * x[j];
#pragma unroll
                for (int i = j + 1; i < K; ++i) s2 = fmaf(L[i][j], x[i], s2);
                z[j] = s2;
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                float s2 = Ld[i] * z[i];
#pragma unroll
                for (int c = 0; c < i; ++c) s2 = fmaf(L[i][c], z[c], s2);
                y[i] = s2 - sigma * x[i];
            }
        };
        auto dot = [&](const float (&x)[K], const float (&y)[K]) {
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < K; ++i) s2 = fmaf(x[i], y[i], s2);
            return s2;
        };
        // lambda_max: power iteration (start vector with all components, not an eigenvector of anything special)
        float x[K], y[K];
#pragma unroll
        for (int i = 0; i < K; ++i) x[i] = 0.2f + 0.05f * (float)((i * 7) % 5);
        float lmax = hi;
        for (int it = 0; it < 12; ++it) {
            amul(x, y);
            const float n2 = dot(y, y);
            const float rn = __builtin_amdgcn_rsqf(n2 > 0.f ? n2 : 1.f);
#pragma unroll
            for (int i = 0; i < K; ++i) x[i] = y[i] * rn;
        }
        amul(x, y);
        lmax = dot(x, y);
        lmax = (lmax > 0.f && lmax <= hi * 1.0001f) ? lmax : hi;
        const bool trunc = !(fabsf(det) >= 1e-4f);
        const float cutoff = 1e-5f * lmax;
        // sub-threshold eigenpairs by deflated inverse iteration
        float v[3][K];
#pragma unroll
        for (int s0 = 0; s0 < 3; ++s0) {
#pragma unroll
            for (int i = 0; i < K; ++i) v[s0][i] = 0.f;     // unused slots must be exact zeros (0 * NaN = NaN)
        }
        int m = 0;
        bool active = trunc && ok;
        bool giveup = !ok;
#pragma unroll
        for (int slot = 0; slot < 4; ++slot) {
            if (!__any(active)) break;
#pragma unroll
            for (int i = 0; i < K; ++i) x[i] = 0.3f + 0.1f * (float)(((i + 3 * slot) * 5) % 7) - 0.05f * (float)slot;
            float lam = 0.f;
            for (int it = 0; it < 6; ++it) {
#pragma unroll
                for (int s0 = 0; s0 < 3; ++s0) {
                    if (s0 < slot) {
                        const float pr = (s0 < m) ? dot(v[s0], x) : 0.f;
#pragma unroll
                        for (int i = 0; i < K; ++i) x[i] = fmaf(-pr, v[s0][i], x[i]);
                    }
                }
                solve(x);
                const float n2 = dot(x, x);
                const float rn = __builtin_amdgcn_rsqf(n2 > 0.f ? n2 : 1.f);
                lam = rn - sigma;                          // 1/||(A+sigma)^-1 x|| -> lambda + sigma
#pragma unroll
                for (int i = 0; i < K; ++i) x[i] *= rn;
            }
            // final clean-up of the accepted vector against the earlier ones
            const bool below = active && (lam <= cutoff);
            if (slot < 3) {
#pragma unroll
                for (int i = 0; i < K; ++i) v[slot][i] = below ? x[i] : 0.f;
                m += below ? 1 : 0;
            } else {
                giveup = giveup || below;              // a 4th sub-threshold eigenvalue: not handled here
            }
            active = below;
        }
        // t = P (A + sigma I)^-1 P w  (+ one refinement step against A when sigma > 0)
        float t[K];
#pragma unroll
        for (int i = 0; i < K; ++i) t[i] = w[i];
        auto project = [&](float (&z)[K]) {
#pragma unroll
            for (int s0 = 0; s0 < 3; ++s0) {
                const float pr = (s0 < m) ? dot(v[s0], z) : 0.f;
#pragma unroll
                for (int i = 0; i < K; ++i) z[i] = fmaf(-pr, v[s0][i], z[i]);
            }
        };
        project(t);
        float wp[K];
#pragma unroll
        for (int i = 0; i < K; ++i) wp[i] = t[i];
        solve(t);
        project(t);
        if (__any(sigma > 0.f)) {
            for (int ref = 0; ref < 2; ++ref) {
                amul(t, y);
#pragma unroll
                for (int i = 0; i < K; ++i) y[i] = wp[i] - y[i];
                project(y);
                solve(y);
                project(y);
#pragma unroll
                for (int i = 0; i < K; ++i) t[i] += (sigma > 0.f) ? y[i] : 0.f;
            }
        }
        // u -= J^T t on the own joint rows; J straight from global (L2-resident: it was just streamed)
        uint32_t fl = (m > 0 ? IRLOSC_FLAG_TRUNCATED : 0u);
        bool bad = false;
        if (live) {
            const float* Jb = p.J + (size_t)b * K * N;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int i = 4 * s + g;
                if (i < N) {
                    float acc = 0.f;
#pragma unroll
                    for (int r = 0; r < K; ++r) acc = fmaf(Jb[r * N + i], t[r], acc);
                    const float uu = p.u[(size_t)b * N + i] - acc;
                    p.u[(size_t)b * N + i] = uu;
                    bad = bad || !t_finite(uu);
                }
            }
        }
        fl |= bad ? IRLOSC_FLAG_NONFINITE : 0u;
        fl = qor(fl);
        if (live && g == 0) {
            p.flags[b] |= fl;
            if (giveup) worklist2[atomicAdd(workcount2, 1)] = b;
        }
    }
}

inline bool group_kernel_supports(int dtype, int n, int k, int ndev) {
    return dtype == IRLOSC_F32 && n == 25 && ((k == 13 && ndev == 3) || (k == 12 && ndev == 2));
}

// Device scratch owned by the context for the two-stage group path.
struct GroupScratch {
    int32_t* worklist;    // [max_batch + 16 * NLISTS] instances flagged by stage 1, NLISTS sharded lists
    int32_t* worklist2;   // [max_batch] instances stage 2 hands to the generic kernel
    int32_t* counts;      // [NLISTS + 1] lengths of the sharded stage-1 lists, then of worklist2
    float* side;          // [(K(K+1)/2 + K)][side_cap]: A and w of flagged instances
    int side_cap;
};

template <typename T>
int launch_group(const KParams<T>& p, const GroupScratch& gs, hipStream_t st);

template <>
inline int launch_group<double>(const KParams<double>&, const GroupScratch&, hipStream_t) {
    return (int)hipErrorNotSupported;
}

template <>
inline int launch_group<float>(const KParams<float>& p, const GroupScratch& gs, hipStream_t st) {
    const int tiles = p.B / grp::TILE;
    const int rem = p.B - tiles * grp::TILE;
    hipError_t e = hipMemsetAsync(gs.counts, 0, (grp::NLISTS + 1) * sizeof(int32_t), st);
    if (e != hipSuccess) return (int)e;
    int32_t* wc1 = gs.counts;
    int32_t* wc2 = gs.counts + grp::NLISTS;
    // each of the NLISTS lists can hold every instance of the tiles that map to it
    const int list_cap = ((tiles + grp::NLISTS - 1) / grp::NLISTS) * grp::TILE;
    if (tiles > 0) {
        int g2 = ((tiles + 3) / 4 + grp::NLISTS - 1) / grp::NLISTS * grp::NLISTS;   // ~1 stage-2 wave per 4 stage-1 waves
        g2 = g2 < grp::NLISTS ? grp::NLISTS : (g2 > 2048 ? 2048 : g2);
        if (p.k == 13 && p.ndev == 3) {
            hipLaunchKernelGGL((osc_group_kernel_f32<13, 3>), dim3(tiles), dim3(64), 0, st, p, gs.worklist, wc1, gs.side, gs.side_cap, list_cap);
            hipLaunchKernelGGL((osc_group_stage2_f32<13>), dim3(g2), dim3(64), 0, st, p, gs.worklist, wc1, gs.side, gs.side_cap, list_cap, gs.worklist2, wc2);
        } else if (p.k == 12 && p.ndev == 2) {
            hipLaunchKernelGGL((osc_group_kernel_f32<12, 2>), dim3(tiles), dim3(64), 0, st, p, gs.worklist, wc1, gs.side, gs.side_cap, list_cap);
            hipLaunchKernelGGL((osc_group_stage2_f32<12>), dim3(g2), dim3(64), 0, st, p, gs.worklist, wc1, gs.side, gs.side_cap, list_cap, gs.worklist2, wc2);
        } else return (int)hipErrorNotSupported;
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    const size_t smem = generic_smem_bytes<float>(p.n, p.k, p.ndev);
    if (rem > 0) {   // ragged tail (< 16 instances): generic kernel on the last instances
        KParams<float> pt = p;
        pt.index = nullptr;
        pt.b0 = tiles * grp::TILE;
        hipLaunchKernelGGL(osc_generic_kernel<float>, dim3(rem), dim3(64), smem, st, pt);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    if (tiles > 0) {
        // instances stage 2 gave up on (> 3 sub-threshold eigenvalues): generic kernel, grid-strided
        KParams<float> pw = p;
        pw.index = gs.worklist2;
        pw.index_count = wc2;
        pw.b0 = 0;
        hipLaunchKernelGGL(osc_generic_kernel<float>, dim3(256), dim3(64), smem, st, pw);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

}  // namespace irlosc
