// Throughput OSC path for the Dual-UR5 shapes (n = 25): FOUR LANES PER ROBOT INSTANCE.
//
// Why not one wavefront per instance: with n = 25 and k <= 13 a 64-lane wave is 60-80 % idle in
// every phase and every operand has to be broadcast, so the VALU issue rate — not HBM — bounds
// the generic kernel at ~5e7 steps/s.  Here a wave carries 16 instances; the 4 lanes of a quad own
// the joint rows i = 4s + g (slot s = 0..6, g = lane & 3; rows 25..27 are zero padding) and keep
// their rows of the Cholesky factor L and of Y = L^-1 J^T in VGPRs with COMPILE-TIME indices
// (everything is fully unrolled).  The only cross-lane traffic is quad-local DPP.
//
// This file: shared helpers (namespace grp), STAGE 2 (stage2_body: the instances whose k x k solve
// is not certifiably the reference's np.linalg.inv branch — deflated inverse iteration applies the
// pinv cut of osc.py:55) and the host-side launch logic.  STAGE 1 (all instances; streaming,
// Cholesky of M, Y, A, certificate, solve, u) is osc_group_stage1.hpp.
//
// Launch structure (launch_group_train): one fused launch = a TRAIN of up to 8 independent steps; in front of each
// step's stage-1 tiles ride the stage-2 blocks of the corresponding step of the PREVIOUS train.  Flagged instances
// hand A and w over in one contiguous record per instance (no worklist); each stage-2 block compacts its own
// 384-instance span in LDS.  Instances stage 2 cannot finish (more than 3 eigenvalues under the cut) go to a small
// give-up list handled by the generic kernel (Jacobi).  profiles/NOTES.md sections 4.2 and 7 have the numbers.
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

// LDS-DMA (global_load_lds_*) is issued through inline asm on purpose: hipcc treats the builtin form as a store to
// LDS that may alias every later ds_read and drains it with s_waitcnt vmcnt(0), which would serialise the ring.  In
// asm the compiler neither counts nor waits for these loads; the kernel's own wait_vm<N>() calls are the only
// synchronisation.  The chunk-level issue code lives in osc_group_stage1.hpp.
__device__ __forceinline__ uint32_t lds_addr(const float* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}

}  // namespace grp

// ---------------------------------------------------------------------------------------------------------
// Second stage for the group kernel: truncated pseudo-inverse solve t = pinv(A, rcond 1e-5) w for the
// instances the first stage could not certify (osc.py:51-55 when |det A| < 1e-4 and cond(A) may exceed
// 1e5), then u -= J^T t.  4 lanes per instance like stage 1 (k x k math replicated in the quad, the
// J^T t update split by joint rows).
//
// Method (A = J M^-1 J^T is symmetric positive semi-definite, k <= 13):
//   * factor A + sigma I = L L^T (sigma = 0, raised to ~2e-6 ||A||_F only if a pivot fails in fp32);
//   * lambda_max by power iteration through the factor (A x = L (L^T x) - sigma x) until its Rayleigh quotient stops growing;
//   * the eigenpairs below the cut 1e-5 lambda_max one at a time by inverse iteration with
//     deflation (at most 3; typically exactly one: a nearly rank-deficient Jacobian stack);
//   * t = P (A + sigma I)^-1 P w with P the projector off those eigenvectors (one refinement step
//     when sigma > 0), which equals sum over the kept eigenpairs of v v^T w / lambda.
// Instances with more than 3 sub-threshold eigenvalues are handed to the generic kernel (Jacobi).
// What stage 2 needs of a step: its J, its outputs and its hand-off buffers.
struct S2Args {
    const float* J;
    float* u;
    uint32_t* flags;
    int nfast;                 // instances handled by stage 1 (multiple of its tile); 0 = nothing pending
    const float* side;
    int side_cap;
    int32_t* worklist2;
    int32_t* workcount2;
};

// Body of stage 2 for block `blk`: ONE LANE PER FLAGGED INSTANCE for the k x k work (it is the same serial
// arithmetic whatever the lane count, so replicating it over a quad only burnt wave slots), then the wave
// cooperates, four lanes per instance, on u -= J^T t.  A block scans S2_SPAN instances and compacts the flagged
// ones in LDS: no global worklist, no atomics.  `lds` = S2_LDS_WORDS 32-bit words.
constexpr int S2_SPAN = 384;     // ~42 flagged per block at the 11 % rate of the synthetic batch: one round of 64
template <int K> struct S2Lds { static constexpr int WORDS = S2_SPAN + 64 * K + 64; };

template <int K>
__device__ __forceinline__ void stage2_body(const S2Args a, int blk, int32_t* lds) {
    using namespace grp;
    constexpr int NA = K * (K + 1) / 2;
    constexpr int SPAN = S2_SPAN;
    int32_t* list = lds;                                         // [SPAN] flagged instance ids
    float* tl = reinterpret_cast<float*>(lds + SPAN);            // [64][K] solutions t of the current round
    uint32_t* fll = reinterpret_cast<uint32_t*>(lds + SPAN + 64 * K);   // [64] their flag bits
    const float* __restrict__ side = a.side;
    const int side_cap = a.side_cap;
    const int lane = threadIdx.x, g = lane & 3, q = lane >> 2;
    const int base = blk * SPAN;
    const unsigned long long below = (1ull << lane) - 1ull;
    int count = 0;
#pragma unroll
    for (int k0 = 0; k0 < SPAN; k0 += 64) {
        const int i0 = base + k0 + lane;
        const bool f0 = i0 < a.nfast && (a.flags[i0] & IRLOSC_FLAG_EIGEN_PATH);
        const unsigned long long m0 = __ballot(f0);
        if (f0) list[count + __popcll(m0 & below)] = i0;
        count += __popcll(m0);
    }
    __syncthreads();
    const int nround = (count + 63) / 64;
    for (int round = 0; round < nround; ++round) {
        const int lpos = round * 64 + lane;
        const bool live = lpos < count;
        const int b = list[live ? lpos : count - 1];
        const int posc = b;                  // the side buffer is indexed by instance
        float L[K][K], Ld[K], Li[K];           // factor: strictly-lower L, diagonal Ld, inverse diagonal Li
        float sigma = 0.f, hi = 0.f, det = 1.f;
        bool ok = false;
        for (int attempt = 0; attempt < 4; ++attempt) {
            float nA2 = 0.f;
            int e = 0;
#pragma unroll
            for (int r = 0; r < K; ++r) {
#pragma unroll
                for (int c = 0; c <= r; ++c) {
                    const float a = side[(size_t)posc * (NA + K) + e];   // this instance's record: one address, immediate offsets
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
        // The Rayleigh quotient of the iterates only grows towards lambda_max: it is run until it stops growing (two
        // consecutive gains under 2e-6, the float32 floor; at most 96 steps), each lane freezing at its own verdict.  Eight
        // steps stood here first: with lambda_2 / lambda_1 ~ 0.85 that is 10-20 % short, the cut comes out too low and an
        // eigenvalue 6-18 % under it is kept -- 30 gross errors (up to 140x) per 65 536 instances of the bench batch
        // (tools/parity_sweep.py --mode f32 --tol 0.1 --band 0.05).
        const bool trunc = !(fabsf(det) >= 1e-4f);
        {
            const float n0 = dot(x, x);
            const float r0 = __builtin_amdgcn_rsqf(n0);
#pragma unroll
            for (int i = 0; i < K; ++i) x[i] *= r0;
        }
        float lmax = 0.f;
        bool pfin = !(trunc && ok);
        int calm = 0;
        for (int it = 0; it < 96; ++it) {
            amul(x, y);
            const float rho = dot(x, y);                       // x is a unit vector
            const float n2 = dot(y, y);
            const float rn = __builtin_amdgcn_rsqf(n2 > 0.f ? n2 : 1.f);
            calm = (it >= 8 && rho - lmax <= 2e-6f * rho) ? calm + 1 : 0;
            lmax = pfin ? lmax : fmaxf(rho, lmax);
            pfin = pfin || calm >= 2;
#pragma unroll
            for (int i = 0; i < K; ++i) x[i] = y[i] * rn;
            if (!__any(!pfin)) break;
        }
        lmax = (lmax > 0.f && lmax <= hi * 1.0001f) ? lmax : hi;
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
            float lam_prev = -1.f;
            // Every lane carries its own instance, so an instance's result must not depend on its wave-mates
            // (sharding a batch differently must not change a single bit): a lane FREEZES x and lambda at its own
            // convergence; the loop merely keeps running until the slowest lane of the wave is done.
            bool fin = !active;
            for (int it = 0; it < 32; ++it) {      // (6 until round 2: a neighbour 25 % above the cut needs ~30 steps to let go)
                float xn[K];
#pragma unroll
                for (int i = 0; i < K; ++i) xn[i] = x[i];
#pragma unroll
                for (int s0 = 0; s0 < 3; ++s0) {
                    if (s0 < slot) {
                        const float pr = (s0 < m) ? dot(v[s0], xn) : 0.f;
#pragma unroll
                        for (int i = 0; i < K; ++i) xn[i] = fmaf(-pr, v[s0][i], xn[i]);
                    }
                }
                solve(xn);
                const float n2 = dot(xn, xn);
                const float rn = __builtin_amdgcn_rsqf(n2 > 0.f ? n2 : 1.f);
                const float lamn = rn - sigma;                 // 1/||(A+sigma)^-1 x|| -> lambda + sigma
                // converged (or clearly above the cut): this lane stops here
                const bool settled = fabsf(lamn - lam_prev) <= 1e-3f * fabsf(lamn) || (it >= 2 && lamn > 4.f * cutoff);
#pragma unroll
                for (int i = 0; i < K; ++i) x[i] = fin ? x[i] : xn[i] * rn;
                lam = fin ? lam : lamn;
                lam_prev = lam;
                fin = fin || (it >= 2 && settled);
                if (!__any(!fin)) break;
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
        float t[K];                            // w is fetched only now: 13 registers less through the iterations
#pragma unroll
        for (int i = 0; i < K; ++i) t[i] = side[(size_t)posc * (NA + K) + NA + i];
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
        // hand t over to the wave: u -= J^T t is done four lanes per instance (coalesced 16-byte groups of J and u)
#pragma unroll
        for (int r = 0; r < K; ++r) tl[lane * K + r] = t[r];
        // the flags must name the branch actually taken here: a failed unshifted factorisation set det = 0, i.e. the
        // reference's pinv branch (osc.py:52-55), even where stage 1's fp32 pivot product had stayed above 1e-4
        fll[lane] = (m > 0 ? IRLOSC_FLAG_TRUNCATED : 0u) | (trunc ? IRLOSC_FLAG_PINV_BRANCH : 0u) | (giveup ? 0x80000000u : 0u);
        __syncthreads();
        const int nlive = (count - round * 64) < 64 ? (count - round * 64) : 64;
        for (int grp4 = 0; grp4 * TILE < nlive; ++grp4) {
            const int slot = grp4 * TILE + q;
            const bool live2 = slot < nlive;
            const int b2 = list[round * 64 + (live2 ? slot : nlive - 1)];
            const float* tq = tl + (live2 ? slot : nlive - 1) * K;
            float t2[K];
#pragma unroll
            for (int r = 0; r < K; ++r) t2[r] = tq[r];
            uint32_t fl = fll[live2 ? slot : nlive - 1];
            bool bad = false;
            if (live2) {
                const float* Jb = a.J + (size_t)b2 * K * N;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const int i = 4 * s + g;
                    if (i < N) {
                        float acc = 0.f;
#pragma unroll
                        for (int r = 0; r < K; ++r) acc = fmaf(Jb[r * N + i], t2[r], acc);
                        const float uu = a.u[(size_t)b2 * N + i] - acc;
                        a.u[(size_t)b2 * N + i] = uu;
                        bad = bad || !t_finite(uu);
                    }
                }
            }
            fl |= bad ? IRLOSC_FLAG_NONFINITE : 0u;
            fl = qor(fl);
            if (live2 && g == 0) {
                if (fl & 0x7fffffffu) a.flags[b2] |= (fl & 0x7fffffffu);
                if (fl & 0x80000000u) a.worklist2[atomicAdd(a.workcount2, 1)] = b2;
            }
        }
        __syncthreads();
    }
}

}  // namespace irlosc
#include "osc_group_stage1.hpp"
namespace irlosc {

inline bool group_kernel_supports(int dtype, int n, int k, int ndev) {
    return dtype == IRLOSC_F32 && n == 25 && ((k == 13 && ndev == 3) || (k == 12 && ndev == 2) || (k == 7 && ndev == 3));
}

constexpr int GROUP_TILE = 16;        // instances per stage-1 wave

}  // namespace irlosc
