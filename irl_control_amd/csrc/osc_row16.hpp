// fp64 throughput kernel: SIXTEEN LANES (one DPP row) PER ROBOT INSTANCE, four instances per wavefront.
//
// The reference computes in float64 end to end (osc.py:49-55,66-67) and north_star's tolerance is 1e-5 relative,
// which the task-space solve only meets in fp64 once cond(J M^-1 J^T) reaches 1e3..1e6.  gfx950 issues v_fma_f64
// at the v_pk_fma_f32 instruction rate, and its only 64-bit DPP mode, row_newbcast (lane N of each 16-lane row to
// the whole row), can be FUSED into v_fmac_f64: "acc += bcast_lane_N(x) * y" is one instruction.  The whole
// algorithm is laid out so that every inner product runs on that instruction and nothing is ever reduced across lanes:
//
//   lane l of a row owns joint rows i = l (slot 0) and i = 16 + l (slot 1, real for l < n - 16) of M and L, and, in a
//   third register set, the right-hand side r = l of the substitution (the "T slot": T[c] = Y[r][c], Y = J L^-T).
//
//   main loop, column j = 0..n-1 (left-looking Cholesky of M with the substitution fused in, osc.py:49-50):
//     m_s  = M[j][i_s]                      the coalesced row j of M IS the column (M symmetric): plain global loads
//     mdq_s += bcast(dq_j) * m_s            uv_all = M dq (osc.py:151)
//     dx    += bcast(dq_j) * J[r][j]        dx = J dq (osc.py:150)
//     for c < j:  m_s -= bcast(L[j][c]) * L[i_s][c];   t -= bcast(L[j][c]) * T[c]     (one DPP source, three uses)
//     d = bcast(m at row j); L[i][j] = m_i / sqrt(d); T[j] = t / sqrt(d)
//   A[r][c] = sum_i bcast(T[i] of lane r) * T[i]   -> lane c holds column c (= row c) of A = J M^-1 J^T, no reduction
//   k x k:  L~ D L~^T of A with column-per-lane storage, W = L~^-1, trace(A^-1) for the condition certificate,
//           t = A^-1 w by W and one back substitution, all on bcast-FMAs                          (osc.py:51-55)
//   u_i  = u0_i + bias_i - kvn * mdq_i - sum_r bcast(t_r) * J[r][i]                               (osc.py:174-200)
//
// Tree form (TOPO = a compiled tree shape; records verified to carry its zeros, or written by the fused walk): M = L^T L with the
// columns taken from the leaves up -- the recursion for hinge j then only involves the hinges of j's own subtree, there is no
// fill-in, and the rows of Y under no end effector are identically zero: 258 column terms instead of 795, 169 products for A
// instead of 325.  Everything else is the same code.
//
// J is read once, coalesced, and parked in LDS (11 KB per wave): column j of it in the right-hand-side layout is one
// conflict-free ds_read per iteration, and the rows come back in the joint layout for J^T t.  Nothing else uses LDS
// beyond a 1 KB exchange area, there is no inter-wave communication and no ring: occupancy is bounded by registers --
// three waves per SIMD (<= 168 registers, no scratch; see IRLOSC_R16_WAVES below).
// Instances whose k x k solve is not certifiably the reference's inverse branch go through the in-wave eigen stage
// (eigen16); the few it gives up on are appended to a worklist and recomputed by the generic kernel (cyclic Jacobi).
//
// Hazard discipline: the hardware does NOT interlock "VALU writes a VGPR -> DPP reads it" (2 wait states; measured:
// tools/probe/dpp64.hip) and the compiler cannot see into inline asm, so (1) every DPP instruction is volatile asm
// (kept in program order), (2) a DPP source is either produced two or more instructions earlier by this file's own
// asm or sits behind a sched_barrier, and (3) the first DPP instruction after a sched_barrier carries "s_nop 1".
#pragma once
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "osc_common.hpp"
#include "osc_frontend.hpp"      // FeCompactTables: the exchange buffer of the fused path (FROMQ)
#include "osc_frontend_lane.hpp" // FeTopo: compile-time queries on a tree shape (FROMQ with a compiled topology)
#include "osc_row16_asm.hpp"     // generated: the main loop's broadcast-FMA chains as asm blocks

// Waves per SIMD the register allocator is asked for, and the prefetch depth of M (rows in flight ahead of the column being
// eliminated).  Three waves need <= 168 registers: every shape compiles to that WITHOUT scratch at depth 4 (at depth 5 and
// more the k = 13 shapes spill, and a kernel that touches scratch at all loses far more than the third wave brings: 175 vs
// 121 us per step).  What made room: the pivot tests pinned where they happen (pinned_ballot), bias / null-space gain /
// instance id fetched where they are used, M dq parked in two free LDS rows across the k x k and eigen stages.
// tools/build_variant.py overrides both for A/B builds.
#ifndef IRLOSC_R16_WAVES
#define IRLOSC_R16_WAVES 3
#endif
#ifndef IRLOSC_R16_PF
#define IRLOSC_R16_PF 4
#endif
#ifndef IRLOSC_EIG_MAXIT       // eigen16: inverse iterations per candidate at most / at least (it >= FLOOR before "settled" counts) / polishing steps
#define IRLOSC_EIG_MAXIT 12
#endif
#ifndef IRLOSC_EIG_FLOOR
#define IRLOSC_EIG_FLOOR 3
#endif
#ifndef IRLOSC_EIG_EXTRA
#define IRLOSC_EIG_EXTRA 2
#endif
#ifndef IRLOSC_R16_TREE_BUDGET          // doubles per lane in flight ahead of the tree form's recursion (dense records)
#define IRLOSC_R16_TREE_BUDGET (2 * IRLOSC_R16_PF)
#endif

namespace irlosc {
namespace r16 {

template <int B_, int E_, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B_ < E_) {
        f(std::integral_constant<int, B_>{});
        static_for<B_ + 1, E_>(f);
    }
}
template <int B_, int E_, typename F>      // E_-1 down to B_
__device__ __forceinline__ void static_for_down(F&& f) {
    if constexpr (B_ < E_) {
        f(std::integral_constant<int, E_ - 1>{});
        static_for_down<B_, E_ - 1>(f);
    }
}

// acc += bcast(src, lane LANE of the row) * mul
template <int LANE>
__device__ __forceinline__ void fmac_bc(double& acc, const double src, const double mul) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(LANE));
}
template <int LANE>
__device__ __forceinline__ void fmac_bc_nop(double& acc, const double src, const double mul) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(LANE));
}
// acc -= bcast(src, LANE) * mul
template <int LANE>
__device__ __forceinline__ void fmac_bc_n(double& acc, const double src, const double mul) {
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(LANE));
}
template <int LANE>
__device__ __forceinline__ void fmac_bc_n_nop(double& acc, const double src, const double mul) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(LANE));
}
// bcast(src, LANE); safe right behind the instruction that wrote src
template <int LANE>
__device__ __forceinline__ double bc_nop(const double src) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(LANE));
    return r;
}

// 32-bit DPP move (the compiler inserts the wait states for builtins)
template <int CTRL>
__device__ __forceinline__ double dpp_mov64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_mov_dpp((int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// sum over the 16 lanes of a row, result in every lane: quad butterflies, then the mirrored half, then the mirrored row
__device__ __forceinline__ double row_sum(double v) {
    v += dpp_mov64<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_mov64<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_mov64<0x141>(v);    // row_half_mirror
    v += dpp_mov64<0x140>(v);    // row_mirror
    return v;
}
__device__ __forceinline__ double quad_bcast(double v, int g) {
    switch (g & 3) {
        case 0: return dpp_mov64<0x00>(v);
        case 1: return dpp_mov64<0x55>(v);
        case 2: return dpp_mov64<0xAA>(v);
        default: return dpp_mov64<0xFF>(v);
    }
}

// 1/sqrt(d) and 1/d to fp64 accuracy from the 2^-24 hardware seeds (one cubic / quadratic correction; measured
// max relative error 2.6e-16 and 0 against 1/sqrt and 1/ of the host, tools/probe/dpp64.hip)
__device__ __forceinline__ double rsq_refined(double d) {
    const double q = __builtin_amdgcn_rsq(d);
    const double e = fma(-(d * q), q, 1.0);
    return fma(q * e, fma(0.375, e, 0.5), q);
}
__device__ __forceinline__ double rcp_refined(double d) {
    const double r = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r, 1.0);
    return fma(r * e, 1.0 + e, r);
}
__device__ __forceinline__ double sqrt_fast(double x) { return x > 0.0 ? x * rsq_refined(x) : 0.0; }

// ---- k x k building blocks: lane c of a row holds column c (= row c) of a symmetric K x K matrix in K registers -----

// In-place L~ D L~^T of A + sigma I.  Out: F[j] in lane c = L~[c][j] for c > j, else 0 (row c of L~); G[i] in lane c =
// L~[i][c] for i > c, else 0 (column c of L~); invd_own = 1 / d_c in lane c; pd = all pivots positive; det = prod d.
// PAD (the KMAX-padded variant of the kernel, see osc_row16_kernel): only the leading kr x kr block is the task space; rows and
// columns kr .. K - 1 of A are exact zeros.  Their pivots are taken as 1 and stay out of `pd` and `det`, their columns of L~ are
// zero: the factorisation of the real block is bit for bit what the K = kr instantiation computes.  The same treatment for the
// rows named in `zrow` (bit j: row j of J is identically zero -- a task row no joint can move, A[j][j] == 0 exactly): such a row
// is an exact null direction of A, decoupled from the rest, and np.linalg.pinv (osc.py:55) drops it whatever the others do.
template <int K, bool PAD = false>
__device__ __forceinline__ void ldl16(double (&A)[K], const int l, const double sigma, double (&F)[K], double (&G)[K],
                                      double& invd_own, bool& pd, double& det, const int kr = K, const uint32_t zrow = 0u) {
    pd = true;
    det = 1.0;
    invd_own = 0.0;
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, K>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        double d = bc_nop<j>(A[j]) + sigma;
        const bool real = !PAD || (j < kr && !((zrow >> j) & 1u));
        const bool npd = real && !(d > 0.0);           // also catches NaN
        pd = pd && !npd;
        d = (npd || !real) ? 1.0 : d;      // the verdict is in: what a non-positive pivot leaves behind is never used (the caller refactors
                                // with a shift or only asked for `pd`); 1 keeps the rest of the recursion finite
        det *= d;
        const double invd = rcp_refined(d);
        const double f = A[j] * invd;
        F[j] = (l > j) ? f : 0.0;
        invd_own = (l == j) ? invd : invd_own;
        __builtin_amdgcn_sched_barrier(0);
        static_for<j + 1, K>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i == j + 1) fmac_bc_n_nop<j>(A[i], A[i], F[j]);
            else fmac_bc_n<j>(A[i], A[i], F[j]);
        });
    });
#pragma unroll
    for (int i = 0; i < K; ++i) G[i] = (l < i) ? A[i] * invd_own : 0.0;
    __builtin_amdgcn_sched_barrier(0);
}

// z <- (L~ D L~^T)^-1 z, one component per lane: two chains of dependent broadcast-FMAs
template <int K>
__device__ __forceinline__ void solve16(double& z, const double (&F)[K], const double (&G)[K], const double invd_own) {
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, K - 1>([&](auto jc) { fmac_bc_n_nop<decltype(jc)::value>(z, z, F[decltype(jc)::value]); });
    z *= invd_own;
    __builtin_amdgcn_sched_barrier(0);
    static_for_down<1, K>([&](auto ic) { fmac_bc_n_nop<decltype(ic)::value>(z, z, G[decltype(ic)::value]); });
    __builtin_amdgcn_sched_barrier(0);
}

// y = A x
template <int K>
__device__ __forceinline__ double matvec16(const double x, const double (&Ac)[K]) {
    double y = 0.0;
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, K>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        if constexpr (r == 0) fmac_bc_nop<r>(y, x, Ac[r]);
        else fmac_bc<r>(y, x, Ac[r]);
    });
    __builtin_amdgcn_sched_barrier(0);
    return y;
}

// t = pinv(A, rcond 1e-5) w for the rows of the wave with `flagged` set (osc.py:55; A symmetric positive semi-definite):
//   the eigenpairs under the "net" 4e-5 ||A||_F one at a time by deflated inverse iteration through the factorisation
//   the caller already has (A + 2^-40 ||A||_F I is factored here instead where that one broke down: exactly singular J);
//   the search stops as soon as trace(A^-1) minus the candidates found proves that nothing else can lie under the net;
//   Rayleigh-Ritz on the span of the candidates when there are several; lambda_max is only bracketed
//   (max diagonal <= lambda_max <= ||A||_F); a candidate inside the bracket of the cut is decided exactly by the inertia
//   of (1e5 th) I - A;
//   t = P A^-1 P w with P the projector off the eigenvectors at or under 1e-5 lambda_max.  Net full: give up (-> Jacobi).
// Every quantity is uniform over the 16 lanes of an instance and frozen at the instance's own convergence, so a
// result never depends on the other instances of the wave (sharding a batch differently changes no bit).
template <int K, bool PAD = false>
__device__ __forceinline__ void eigen16(const double (&Ac)[K], double (&F)[K], double (&G)[K], double& invd_own, const bool pdA,
                                        const double nA2, const double trA, const double w, const int l, const bool flagged,
                                        double& t, uint32_t& fl, bool& giveup, const int kr = K, const uint32_t zrow = 0u) {
    const double hi = sqrt(nA2);
    double sigma = 0.0;
    giveup = flagged && !(hi > 0.0 && t_finite(hi));
    // The plain factorisation is unusable when a pivot is non-positive -- or positive at rounding level: J singular to working
    // precision leaves a pivot of +-1e-17 ||A||, and the sign is luck (tools/parity_sweep.py --stress found the positive ones:
    // errors up to 200x).  trace(A^-1) ||A||_F >= cond(A) tells: beyond 1e11 the factorisation is redone on A + sigma.
    const bool broken = !pdA || !(trA * hi < 1e11);
    if (__any(flagged && broken)) {
        // The wave factors again, that instance with sigma = 2^-40 ||A||_F on the diagonal (cond <= 1.1e12; the retained
        // eigenvalues, >= 1e-5 lambda_max, move by <= 3e-7 relative), the others with sigma = 0 (which reproduces their factors).
        double A2[K], det2;
        bool pd2;
#pragma unroll
        for (int r = 0; r < K; ++r) A2[r] = Ac[r];
        const bool use = flagged && broken;
        sigma = use ? hi * 0x1p-40 : 0.0;
        ldl16<K, PAD>(A2, l, sigma, F, G, invd_own, pd2, det2, kr, zrow);
        giveup = giveup || (use && !pd2);
    }
    // bracket of lambda_max: the largest diagonal entry is a Rayleigh quotient, the Frobenius norm an upper bound
    double diag = 0.0;
#pragma unroll
    for (int r = 0; r < K; ++r) diag = (l == r) ? Ac[r] : diag;
    double lo = diag;
    lo = fmax(lo, dpp_mov64<0xB1>(lo)); lo = fmax(lo, dpp_mov64<0x4E>(lo));
    lo = fmax(lo, dpp_mov64<0x141>(lo)); lo = fmax(lo, dpp_mov64<0x140>(lo));
    lo = (lo > 0.0 && lo <= hi) ? lo : hi * 0.25;
    const double net = 4e-5 * hi;
    // Candidate eigenpairs: everything under the net, one at a time by deflated inverse iteration.  The net is wider than
    // the cut on purpose: two eigenvalues straddling the cut converge towards each other's mixtures (ratio close to 1),
    // but the SPAN of the candidates converges at the rate of the gap to the first eigenvalue outside the net (<= 1/4
    // per iteration), and the Rayleigh-Ritz step below then separates them exactly.
    constexpr int NV = 4;
    double v[NV] = {0.0, 0.0, 0.0, 0.0};
    double th[NV] = {0.0, 0.0, 0.0, 0.0};      // Ritz value of v[i]
    bool has[NV] = {false, false, false, false};
    int m = 0;
    bool active = flagged && !giveup;
    double rem = (pdA && sigma == 0.0 && trA > 0.0) ? trA : -1.0;      // trace(A^-1) minus 1/mu of the candidates found
#pragma unroll
    for (int slot = 0; slot < NV; ++slot) {
        if (!__any(active)) break;
        int lq = l;
        asm volatile("" : "+v"(lq));       // (keeps the four start vectors from being computed ahead of the stage and carried through it)
        const bool inblock = PAD ? (lq < kr && !((zrow >> lq) & 1u)) : lq < K;      // (zero outside the real block, and it stays zero: A is block diagonal)
        double x = inblock ? 0.3 + 0.1 * (double)(((lq + 3 * slot) * 5) % 7) - 0.05 * (double)slot : 0.0;
        double lam = 0.0, lam_prev = -1.0;
        bool fin = !active;
        for (int it = 0; it < IRLOSC_EIG_MAXIT; ++it) {
            double xn = x;
#pragma unroll
            for (int s0 = 0; s0 < NV - 1; ++s0) {
                if (s0 < slot) xn = fma(-row_sum(v[s0] * xn), v[s0], xn);     // unused v are exact zeros
            }
            solve16<K>(xn, F, G, invd_own);
            const double n2 = row_sum(xn * xn);
            const double rn = rsq_refined(n2 > 0.0 ? n2 : 1.0);
            const double lamn = rn - sigma;                 // 1 / ||(A + sigma)^-1 x|| -> lambda + sigma (from above)
            const bool settled = fabs(lamn - lam_prev) <= 1e-10 * fabs(lamn) || lamn > 4.0 * net;
            x = fin ? x : xn * rn;
            lam = fin ? lam : lamn;
            lam_prev = lam;
            fin = fin || (it >= IRLOSC_EIG_FLOOR && settled);
            if (!__any(!fin)) break;
        }
        // The eigenvalue is settled to 1e-10, which leaves the VECTOR at ~1e-5 -- and what leaks past the projector is
        // amplified by 1 / lambda_cut: two more steps (each gains at least the factor 4 of the net, typically far more) for
        // every instance, whenever its loop froze (tools/parity_sweep.py --stress --layout k7: errors of 1.2e-5 .. 1.6e-5).
#pragma unroll
        for (int ex = 0; ex < IRLOSC_EIG_EXTRA; ++ex) {
            double xn = x;
#pragma unroll
            for (int s0 = 0; s0 < NV - 1; ++s0) {
                if (s0 < slot) xn = fma(-row_sum(v[s0] * xn), v[s0], xn);
            }
            solve16<K>(xn, F, G, invd_own);
            const double n2 = row_sum(xn * xn);
            const double rn = rsq_refined(n2 > 0.0 ? n2 : 1.0);
            const bool ok = n2 > 0.0 && t_finite(rn);
            x = ok ? xn * rn : x;
            lam = (ok && lam <= 4.0 * net) ? rn - sigma : lam;
        }
        const bool cand = active && (lam <= net);
        // final clean-up against the earlier vectors (the last solve re-introduced rounding-level components)
#pragma unroll
        for (int s0 = 0; s0 < NV - 1; ++s0) {
            if (s0 < slot) x = fma(-row_sum(v[s0] * x), v[s0], x);
        }
        {
            const double n2 = row_sum(x * x);
            x *= rsq_refined(n2 > 0.0 ? n2 : 1.0);
        }
        v[slot] = cand ? x : 0.0;
        th[slot] = cand ? lam : 0.0;
        has[slot] = cand;
        m += cand ? 1 : 0;
        // what is left of trace(A^-1) bounds the next eigenvalue from below: lambda_next >= 1 / rem.  If that clears the
        // net there is nothing more to find.  The candidate's eigenvalue is settled to 1e-10 relative, so the difference is
        // trusted while it keeps 1e-7 of the minuend (three digits of margin).
        rem = (cand && rem > 0.0 && lam > 0.0) ? rem - rcp_refined(lam) : (cand ? -1.0 : rem);
        const bool exhausted = rem > 1e-7 * trA && rem * net < 1.0;
        active = cand && !exhausted;
    }
    giveup = giveup || (m == NV);               // the net is full: there may be more under it (-> Jacobi)
    // Rayleigh-Ritz on the span of the candidates whenever an instance has more than one: H = V^T A V (4 x 4, the
    // unused vectors are zero), cyclic Jacobi on H with the rotations applied to V.  Rare, wave-uniform branch.
    if (__any(m >= 2)) {
        double av[NV], h[NV][NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) av[i] = matvec16<K>(v[i], Ac);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
#pragma unroll
            for (int j = i; j < NV; ++j) { h[i][j] = row_sum(v[i] * av[j]); h[j][i] = h[i][j]; }
        }
        for (int sweep = 0; sweep < 6; ++sweep) {
            bool turned = false;               // a sweep without a rotation anywhere in the wave: all later ones are identities
#pragma unroll
            for (int p2 = 0; p2 < NV - 1; ++p2) {
#pragma unroll
                for (int q2 = p2 + 1; q2 < NV; ++q2) {
                    const double hpq = h[p2][q2], hpp = h[p2][p2], hqq = h[q2][q2];
                    const bool rot = has[p2] && has[q2] && fabs(hpq) > 1e-300 && fabs(hpq) > 1e-18 * (fabs(hpp) + fabs(hqq));
                    turned = turned || rot;
                    double theta = (hqq - hpp) * rcp_refined(rot ? 2.0 * hpq : 1.0);
                    theta = fmin(fmax(theta, -1e100), 1e100);      // theta^2 must stay finite (then t ~ 1 / (2 theta) ~ 0)
                    const double tq = (theta >= 0.0 ? 1.0 : -1.0) * rcp_refined(fabs(theta) + sqrt_fast(theta * theta + 1.0));
                    const double cs = rot ? rsq_refined(tq * tq + 1.0) : 1.0;
                    const double sn = rot ? tq * cs : 0.0;
                    h[p2][p2] = rot ? hpp - tq * hpq : hpp;
                    h[q2][q2] = rot ? hqq + tq * hpq : hqq;
                    h[p2][q2] = rot ? 0.0 : hpq;
                    h[q2][p2] = h[p2][q2];
#pragma unroll
                    for (int r = 0; r < NV; ++r) {
                        if (r != p2 && r != q2) {
                            const double hrp = h[r][p2], hrq = h[r][q2];
                            h[r][p2] = cs * hrp - sn * hrq; h[p2][r] = h[r][p2];
                            h[r][q2] = sn * hrp + cs * hrq; h[q2][r] = h[r][q2];
                        }
                    }
                    const double vp = v[p2], vq = v[q2];
                    v[p2] = cs * vp - sn * vq;
                    v[q2] = sn * vp + cs * vq;
                }
            }
            if (!__any(turned)) break;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) th[i] = (m >= 2 && has[i]) ? h[i][i] - 0.0 : th[i];
    }
    // The pinv cut (osc.py:55): drop the Ritz pairs at or under 1e-5 lambda_max.  lambda_max itself is never computed: it is
    // bracketed by [lo, hi], a value at or under 1e-5 lo is cut and one over 1e-5 hi is kept whatever it is, and for a value
    // th in between the question "th <= 1e-5 lambda_max" is "lambda_max >= beta = 1e5 th", i.e. "beta I - A is NOT positive
    // definite" (Sylvester's law of inertia): one more L D L^T in the layout at hand, exact to rounding.  (Two estimates
    // stood here before -- a 24-step power iteration with a +-2 % ambiguity window, then a growing Rayleigh quotient with
    // a geometric extrapolation of what it could still gain -- and tools/parity_sweep.py broke both: lambda_2 / lambda_1 =
    // 0.85 leaves the first 6 % short, a mix of fast and slow modes fools the second.)  Every instance asks about its own
    // values only, so nothing depends on its wave-mates.
    int ncut = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const bool below = has[i] && th[i] <= 1e-5 * lo;
        const bool ask = has[i] && !below && th[i] <= 1e-5 * hi;
        bool cut = below;
        if (__any(ask)) {
            double A2[K], Ft[K], Gt[K], invt = 0.0, dett = 1.0;
            bool pdt = true;
#pragma unroll
            for (int r = 0; r < K; ++r) A2[r] = -Ac[r];
            ldl16<K, PAD>(A2, l, ask ? th[i] * 1e5 : 4.0 * hi, Ft, Gt, invt, pdt, dett, kr, zrow);
            cut = cut || (ask && !pdt);
        }
        v[i] = cut ? v[i] : 0.0;
        ncut += cut ? 1 : 0;
    }
    // t = P (A + sigma)^-1 P w
    double tt = w;
    bool live[NV];                     // a vector that is zero in every lane of the wave projects nothing (the usual case:
#pragma unroll                         // one eigenvector cut, or none)
    for (int s0 = 0; s0 < NV; ++s0) live[s0] = __any(v[s0] != 0.0);
#pragma unroll
    for (int s0 = 0; s0 < NV; ++s0)
        if (live[s0]) tt = fma(-row_sum(v[s0] * tt), v[s0], tt);
    solve16<K>(tt, F, G, invd_own);
#pragma unroll
    for (int s0 = 0; s0 < NV; ++s0)
        if (live[s0]) tt = fma(-row_sum(v[s0] * tt), v[s0], tt);
    t = tt;
    fl = ncut > 0 ? IRLOSC_FLAG_TRUNCATED : 0u;
}

// Velocity limiting + gains + stiffness (osc.py:70-99,160-168) like apply_gains6 (osc_common.hpp), with the divisions
// and square roots on the refined hardware seeds (a few ulp from the IEEE results; the IEEE sequences cost 15-25
// instructions each and this runs on every lane).
__device__ __forceinline__ void apply_gains6_fast(const double* __restrict__ g, double e[6]) {
    const double kp = g[0], kv = g[1], ko = g[2];
    if (g[11] != 0.0) {
        double sx = 1.0, sa = 1.0;
        const double rkv = rcp_refined(kv);
        const double nx = sqrt_fast(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        const double satx = g[9] * rcp_refined(kp) * kv;
        if (nx > satx) sx = satx * rcp_refined(nx);
        const double na = sqrt_fast(e[3] * e[3] + e[4] * e[4] + e[5] * e[5]);
        const double sata = g[10] * rcp_refined(ko) * kv;
        if (na > sata) sa = sata * rcp_refined(na);
        const double lp = kp * rkv, lo = ko * rkv;  // lamb (osc.py:39)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            e[i] = kv * sx * lp * e[i] * g[3 + i];  // stiffness k; abg entries are 1
            e[3 + i] = kv * sa * lo * e[3 + i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            e[i] *= kp * g[3 + i];
            e[3 + i] *= ko;
        }
    }
}

// Orientation part of calc_error (osc.py:113-117): the entries of R = quat2mat(qconjugate(qmult(normalized(q_tgt), qconjugate(q_ee))))
// that mat2euler('sxyz') reads, with cy and the gimbal-lock verdict -- same formulas as task_error6 (osc_common.hpp), divisions and
// square roots on the refined hardware seeds.  The three angles are atan2(ay, ax) of TaskRot::angle_args.
struct TaskRot {
    double r00, r10, r20, r21, r22, r11, r12, cy;
    bool gimbal;
    __device__ __forceinline__ void angle_args(const int a, double& ay, double& ax) const {
        ay = 0.0; ax = 1.0;                        // atan2(0, 1) = 0: an idle lane, and az at gimbal lock
        if (a == 0) { ay = gimbal ? -r12 : r21; ax = gimbal ? r11 : r22; }
        else if (a == 1) { ay = -r20; ax = cy; }
        else if (a == 2 && !gimbal) { ay = r10; ax = r00; }
    }
};
__device__ __forceinline__ TaskRot task_rot(const double (&ee)[7], const double (&tg)[7]) {
    const double tw = tg[3], tx = tg[4], ty = tg[5], tz = tg[6];
    const double rnrm = rsq_refined(tw * tw + tx * tx + ty * ty + tz * tz);
    const double w1 = tw * rnrm, x1 = tx * rnrm, y1 = ty * rnrm, z1 = tz * rnrm;
    const double w2 = ee[3], x2 = -ee[4], y2 = -ee[5], z2 = -ee[6];
    const double rw = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2;
    const double rx = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2;
    const double ry = w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2;
    const double rz = w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2;
    const double w = rw, xq = -rx, yq = -ry, zq = -rz;
    const double Nq = w * w + xq * xq + yq * yq + zq * zq;
    TaskRot R{1.0, 0.0, 0.0, 0.0, 1.0, 1.0, 0.0, 0.0, false};
    if (!(Nq < 2.220446049250313e-16)) {
        const double s = 2.0 * rcp_refined(Nq);
        const double X = xq * s, Y = yq * s, Z = zq * s;
        const double wX = w * X, wY = w * Y, wZ = w * Z;
        const double xX = xq * X, xY = xq * Y, xZ = xq * Z;
        const double yY = yq * Y, yZ = yq * Z, zZ = zq * Z;
        R.r00 = 1.0 - (yY + zZ); R.r10 = xY + wZ; R.r20 = xZ - wY; R.r21 = yZ + wX;
        R.r22 = 1.0 - (xX + yY); R.r11 = 1.0 - (xX + zZ); R.r12 = yZ - wX;
    }
    R.cy = sqrt_fast(R.r00 * R.r00 + R.r10 * R.r10);
    R.gimbal = !(R.cy > 4.0 * 2.220446049250313e-16);
    return R;
}

// Compile-time shape queries for the tree-structured factorisation (hinge numbering = MuJoCo's depth-first order)
template <class TOPO>
constexpr int tree_subtree_size(int j) {
    int n = 0;
    for (int c = 0; c < TOPO::NJ; ++c) n += FeTopo<TOPO>::above(j, c) ? 1 : 0;
    return n;
}
template <class TOPO>
constexpr bool tree_subtree_contiguous(int j) {
    const int sz = tree_subtree_size<TOPO>(j);
    for (int c = 0; c < TOPO::NJ; ++c)
        if (FeTopo<TOPO>::above(j, c) != (c >= j && c < j + sz)) return false;
    return true;
}

// hinge j moves a body that may be named as an end effector: every other column of J is structurally zero, and with the
// factorisation running from the leaves up (row j of Y = L^-T J^T only sees rows of its own subtree) so is that row of Y
template <class TOPO>
constexpr bool tree_moves_ee(int j) {
    for (int b = 0; b < TOPO::NB; ++b)
        if (TOPO::ee_cand[b] != 0 && FeTopo<TOPO>::moves(j, b)) return true;
    return false;
}
template <class TOPO>
constexpr int tree_ee_run(int c0, int cend) {          // length of the run of columns from c0 that are all EE hinges / all not
    int n = 1;
    while (c0 + n < cend && tree_moves_ee<TOPO>(c0 + n) == tree_moves_ee<TOPO>(c0)) ++n;
    return n;
}
// row j of M has no structural non-zero in columns 16 .. NJ - 1 (its slot-1 half): nothing to load, nothing to multiply
template <class TOPO>
constexpr bool tree_slot1_zero(int j) {
    for (int c = 16; c < TOPO::NJ; ++c)
        if (FeTopo<TOPO>::above(j, c) || FeTopo<TOPO>::above(c, j)) return false;
    return true;
}
// Prefetch plan of the tree form on dense records: rows are requested in processing order (NJ - 1 .. 0) so that at most
// BUDGET doubles per lane are in flight -- a row with an empty slot-1 half costs one register pair instead of two, so twice
// as many of those fit.  Entry 0 = the prologue, entry i + 1 = what step i requests once it has taken its own row;
// first / count are POSITIONS in the processing order (position p = row NJ - 1 - p).
template <class TOPO, int BUDGET>
struct TreeFetch {
    int first[TOPO::NJ + 1];
    int count[TOPO::NJ + 1];
};
template <class TOPO, int BUDGET>
constexpr TreeFetch<TOPO, BUDGET> tree_fetch() {
    constexpr int NJ = TOPO::NJ;
    static_assert(BUDGET >= 2, "a full row must fit");
    TreeFetch<TOPO, BUDGET> f{};
    int next = 0, inflight = 0;
    for (int e = 0; e <= NJ; ++e) {
        if (e > 0) inflight -= tree_slot1_zero<TOPO>(NJ - e) ? 1 : 2;      // step e - 1 has taken row NJ - 1 - (e - 1)
        f.first[e] = next;
        int n = 0;
        while (next < NJ && inflight + (tree_slot1_zero<TOPO>(NJ - 1 - next) ? 1 : 2) <= BUDGET) {
            inflight += tree_slot1_zero<TOPO>(NJ - 1 - next) ? 1 : 2;
            ++next;
            ++n;
        }
        f.count[e] = n;
    }
    return f;
}
template <class TOPO, int BUDGET>
inline constexpr TreeFetch<TOPO, BUDGET> kTreeFetch = tree_fetch<TOPO, BUDGET>();

// bit c of word j: M[j][c] may be non-zero (c at or above j, or j above c); bit c of *jcols: column c of J may be non-zero
template <class TOPO>
inline void tree_structure_masks(uint32_t mrow[32], uint32_t* jcols) {
    *jcols = 0;
    for (int j = 0; j < 32; ++j) {
        mrow[j] = 0;
        if (j >= TOPO::NJ) continue;
        for (int c = 0; c < TOPO::NJ; ++c)
            if (FeTopo<TOPO>::above(j, c) || FeTopo<TOPO>::above(c, j)) mrow[j] |= 1u << c;
        if (tree_moves_ee<TOPO>(j)) *jcols |= 1u << j;
    }
}

// The lanes where `c` holds, as a scalar mask, evaluated HERE: without the pin the compiler sinks the 25 pivot tests of the
// main loop to where `flags` is finally used and carries all 25 pivots there in vector registers.
__device__ __forceinline__ unsigned long long pinned_ballot(const bool c) {
    unsigned long long m = __ballot(c);
    asm volatile("" : "+s"(m));
    return m;
}

// LDS hand-over inside ONE wave (64-thread blocks): DS operations of a wave execute in order, so a compile-time
// ordering point plus "all my DS operations are done" is a complete synchronisation.  __syncthreads() would also drain
// every global load in flight (its fence covers all address spaces: s_waitcnt vmcnt(0)), i.e. the prefetched M stream.
__device__ __forceinline__ void lds_sync() {
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xF | (0x7 << 4) | (0x0 << 8) | (0x3 << 14));      // lgkmcnt(0), vmcnt / expcnt untouched
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace r16

// What the row16 kernel needs beyond KParams: a page of zeros (padding lanes load from it instead of being masked),
// and the worklist of the instances handed to the generic kernel.
struct Row16Extra {
    const void* zeros;         // >= 32 * 32 * 8 bytes of zeros
    int32_t* worklist;         // [B]
    int32_t* workcount;        // counter of this step (zero on entry)
    // FROMQ (the fused path from joint coordinates): M, J, bias and the end-effector poses are not records in HBM but
    // entries of the compact exchange buffer the lane-per-robot walk left behind (osc_frontend_lane.hpp), dq is qvel
    const double* side;        // [walk wave][entry][64 robots]
    const double* qvel;        // [B][n]
    const FeCompactTables* tables;
    // irlosc_time_trains: {min over waves of the start, max over waves of the end} of the 100 MHz wall clock (s_memrealtime),
    // R16_SPAN_SLOTS pairs per TRAIN (every step of a train points at the same block; a wave uses pair blockIdx.x % slots: 131 072
    // waves hammering ONE address serialise in the L2 -- measured: a train took 3.07 ms instead of 0.85); nullptr = no stamps
    unsigned long long* span;
    // dense records: part 1 of the task-space signal as k rows per instance, [B][16] doubles, left by osc_task_rows_dense_kernel
    // ahead of this kernel; nullptr: computed in the kernel
    const double* trows;
};

// One launch = a TRAIN of up to R16_TRAIN steps (blockIdx.y = step): consecutive steps of irlosc_step_resident are
// independent batches (different resident slots, different output sets), so chaining them in one grid removes the
// launch gap, the ramp-up and the tail between them (a step's last waves - those with eigen-stage instances - finish
// while the next step's first waves already run).  A train of one step is the plain single-step launch.
constexpr int R16_TRAIN = 8;
constexpr int R16_SPAN_SLOTS = 256;      // stamp pairs per train (irlosc_time_trains), a power of two; + one pair {shader cycles, wall-clock ticks} of a sample wave
constexpr int R16_SPAN_WORDS = 2 * R16_SPAN_SLOTS + 2;
template <typename TIN>
struct Row16Train {
    KParams<TIN> p[R16_TRAIN];
    Row16Extra x[R16_TRAIN];
};

// TIN = storage type of the records (double, or float for the mixed path); arithmetic is double throughout.
// FROMQ: the fused path from joint coordinates.  Same kernel, but the operands the rigid-body front end computes come out
// of the compact exchange buffer (x.side, entry tables x.tables) instead of dense records: row j of M is 25 gathered
// entries of which the structural zeros all point at one entry of zeros, J likewise, bias and EE pose are entries, dq is
// qvel.  A walk wave's block [entry][64 robots] is consumed by 16 blocks of this kernel (4 robots each); the
// blockIdx -> robots map keeps those 16 on ONE XCD (blockIdx.x % 8), so that the 128-byte lines they share are fetched
// into one L2 only.  Targets, gains, wrench, outputs stay records of type TIN.
//
// TOPO (FROMQ only): the compiled tree shape of the model.  The joint-space inertia of a TREE factors without fill-in when
// the hinges are eliminated leaves first: M = L^T L with L[c][i] != 0 only for i at or above c.  In MuJoCo's depth-first
// numbering that is the Cholesky recursion run from the LAST column to the first, and column j only needs the terms
// c in subtree(j) \ {j} = j + 1 .. j + size(j) - 1, a contiguous run known at compile time: for the Dual-UR5 130 column terms
// instead of the 300 of a dense 25 x 25 factorisation, and two instead of three chains for the columns below 16 (their
// slot-1 rows are already eliminated) -- 283 broadcast-FMAs per wave instead of 745.  The entries of unrelated pairs come out
// as exact zeros (0 - sum of products with exact zeros, scaled), so nothing has to be masked; Y = L^-T J^T rides along as
// before and J M^-1 J^T = Y^T Y is the same identity.  Records of unknown origin (TOPO = void) keep the dense recursion.
template <class TOPO>
constexpr bool tree_row_slot1_zero(int j) {
    if constexpr (std::is_void_v<TOPO>) return false;
    else return r16::tree_slot1_zero<TOPO>(j);
}
template <class TOPO>
constexpr bool tree_row_of_y(int i) {
    if constexpr (std::is_void_v<TOPO>) return true;
    else return r16::tree_moves_ee<TOPO>(i);
}
// Column terms c = C0 .. C0 + NC - 1 (the subtree of hinge J below it) of the tree-structured recursion: the rows of the
// factor in R0 / R1, of Y in T; runs of columns under an end effector carry the Y chain, the others (T[c] = 0) do not.
template <class TOPO, int J, int C0, int NC, int NR0, int NR1, int NT>
__device__ __forceinline__ void tree_chains(double& m0, double& m1, double& tj, const double (&R0)[NR0], const double (&R1)[NR1],
                                            const double (&T)[NT]) {
    if constexpr (NC > 0) {
        constexpr int gj = J & 15;
        constexpr int RUN = r16::tree_ee_run<TOPO>(C0, C0 + NC);
        constexpr bool y = r16::tree_moves_ee<TOPO>(J) && r16::tree_moves_ee<TOPO>(C0);
        if constexpr (J >= 16) {                                  // row J is a slot-1 row: the broadcasts come from R1
            if constexpr (y) r16::fmac3_chain<gj, C0, RUN>(m1, m0, tj, R1, R0, T);
            else r16::fmac2_chain<gj, C0, RUN>(m1, m0, R1, R0);
        } else {                                                  // slot-1 rows are done
            if constexpr (y) r16::fmac2_chain<gj, C0, RUN>(m0, tj, R0, T);
            else r16::static_for<C0, C0 + RUN>([&](auto cc) { r16::fmac_bc_n<gj>(m0, R0[decltype(cc)::value], R0[decltype(cc)::value]); });
        }
        tree_chains<TOPO, J, C0 + RUN, NC - RUN>(m0, m1, tj, R0, R1, T);
    }
}

//
// PAD: the KMAX-padded variant for every layout without an instantiation of its own (osc.py:134-138 stacks J over WHATEVER targets
// are passed, examples/ps_move_example.py:137-150 re-masks a device between ticks): K is then an upper bound KMAX >= p.k, NDEV is
// IRLOSC_MAX_DEV, and the real k = p.k and ndev = p.ndev are kernel arguments (scalar registers).  Rows k .. KMAX - 1 of J are loaded
// from the page of zeros, so Y, A = Y^T Y, w and t carry exact zeros there; the pivots of the padded block are taken as 1 and kept out
// of det, trace(A^-1), ||A||_F and the inertia counts (ldl16), the start vectors of the inverse iteration are zero there (and stay
// zero: A is block diagonal), so det, lambda_max, the 1e-5 cut and every flag are those of the k x k problem -- and since padding only
// ever adds exact zeros to sums, the torques equal those of a K = k instantiation bit for bit.
template <int K, int NDEV, typename TIN, int N, bool FROMQ = false, class TOPO = void, bool PAD = false>
__global__ __launch_bounds__(FROMQ ? 256 : 64, K > 13 ? 2 : IRLOSC_R16_WAVES)      // (K = 14 .. 16 spills at three waves per SIMD: two)
void osc_row16_kernel(const Row16Train<TIN> tr) {
    using namespace r16;
    static_assert(!PAD || NDEV == IRLOSC_MAX_DEV, "the padded variant takes the number of devices at run time");
    constexpr bool TREE = !std::is_void_v<TOPO>;      // the records carry the zero pattern of this tree (fused path: by
                                                      // construction; dense records: verified when they were uploaded)
    static_assert(!FROMQ || TREE, "the fused path exists for a compiled tree shape");
    // Dense records: one wave per block (4 instances).  FROMQ: FOUR waves per block = 16 robots = one 128-byte line of every
    // entry of the walk's exchange block [entry][64 robots]; the block stages those lines in LDS once (r16 tile, below).
    constexpr int NW = FROMQ ? 4 : 1;
    const KParams<TIN>& p = tr.p[blockIdx.y];
    const Row16Extra& x = tr.x[blockIdx.y];
    const int kr = PAD ? p.k : K;              // real task rows / target devices (scalar; compile-time constants unless PAD)
    const int nd = PAD ? p.ndev : NDEV;
    using TM = std::conditional_t<FROMQ, double, TIN>;      // type the M / J / dq / bias operands arrive in
    static_assert(N > 16 && N <= 32 && K >= 1 && K <= 16 && NDEV >= 1 && NDEV <= 4, "shape");
    constexpr int N1 = N - 16;                 // real rows in slot 1
    constexpr int PF = IRLOSC_R16_PF;          // rows of M in flight ahead of the column being eliminated
    __shared__ double Jl[FROMQ ? 2 : 4 * (K + 1) * N + 16];   // dense records: [q][r][i]; row K of each instance is zeros (lanes >= K read it)
    __shared__ double Wl_[NW][4][16];          // task vector, written by the device lanes
    __shared__ double Dxl_[NW][4][16];         // dx for the target-velocity branch
    __shared__ double Kvl_[NW][4][4];
    __shared__ int Brl_[NW][4][4];
    // FROMQ: the block's tile of the exchange buffer, tile[entry][16 robots + 1 pad] (43 KB: three blocks = twelve waves per
    // CU), and the entry tables as byte offsets into it (FeCompactTables::t_m .. t_b, one contiguous block of uint16)
    constexpr int TILE_E = [] { if constexpr (FROMQ) return FeTopo<TOPO>::task_index(0) + K; else return 1; }();     // rows staged
    constexpr int BLK_E = [] { if constexpr (FROMQ) return FeTopo<TOPO>::n_compact(); else return 1; }();            // rows of a walk block
    __shared__ double Tile[FROMQ ? TILE_E * FE_TILE_ROW : 2];
    __shared__ __align__(16) uint16_t Tab[FROMQ ? FE_TILE_TAB_WORDS : 8];
    const uint16_t* const Tm = Tab;                                 // [j * 32 + i]: M[i][j]
    const uint16_t* const Tj = Tab + 32 * 32;                       // [r * 32 + i]: J[r][i]
    const uint16_t* const Tee = Tab + 32 * 32 + IRLOSC_MAX_K * 32;  // [d * 8 + c]
    const uint16_t* const Tb = Tee + IRLOSC_MAX_DEV * 8;            // [i]
    const uint16_t* const Te = Tb + 32;                             // [r]: gained task error, row r
    const int wv = FROMQ ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;      // wave of the block (scalar)
    const int lane = threadIdx.x & 63, q = lane >> 4, l = lane & 15;
    double (&Wl)[4][16] = Wl_[wv];
    double (&Dxl)[4][16] = Dxl_[wv];
    double (&Kvl)[4][4] = Kvl_[wv];
    int (&Brl)[4][4] = Brl_[wv];
    const int blk = blockIdx.x;
    const int b = blk * (4 * NW) + (int)(threadIdx.x >> 4);
    const bool live = b < p.B;
    const int bc = live ? b : p.B - 1;
    const bool v1 = l < N1;
    const TIN* __restrict__ zeros = reinterpret_cast<const TIN*>(x.zeros);
    const TIN* __restrict__ m0p = FROMQ ? zeros : p.M + (size_t)bc * (N * N) + l;
    const TIN* __restrict__ m1p = (!FROMQ && v1) ? p.M + (size_t)bc * (N * N) + 16 + l : zeros;
    // FROMQ: the 16 robots of a block sit in ONE walk wave's exchange block (block x -> walk wave x / 4, robots 16 (x % 4) ..):
    // all four waves pull the block's 128-byte lines -- entry e, 16 robots -- into the tile, every line exactly once, perfectly
    // coalesced (a wave load = 4 whole lines); the main loop then gathers out of LDS.  (Round 3 gathered from L2 in the main
    // loop: a row of M was 16 different lines of which 32 bytes each were used, 1 165 us per train against 968 for dense records.)
    const unsigned rob8 = (threadIdx.x >> 4) * 8u;                  // this lane's robot: byte offset inside a tile row
    auto tile_at = [&](const unsigned row_off) -> double {          // row_off = entry x FE_TILE_ROW_BYTES (from the tables)
        return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(Tile) + (row_off + rob8));
    };
    // IRLOSC_PHASE_TIMING=1 debug runs: cycle stamps per phase and the wall clock of the wave (p.dbg != nullptr);
    // FROMQ: the tile fill below belongs to the first phase
    unsigned long long ts[8];
    const unsigned long long rt0 = (p.dbg || x.span) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const unsigned long long cyc0 = x.span ? (unsigned long long)__builtin_readcyclecounter() : 0ull;
#define IRLOSC_TS(i) ts[i] = p.dbg ? __builtin_readcyclecounter() : 0ull
    IRLOSC_TS(0);
    if constexpr (FROMQ) {
        const FeCompactTables* __restrict__ tb = x.tables;
        const double* __restrict__ src = x.side + ((size_t)(blk >> 2) * BLK_E) * 64 + (blk & 3) * 16;
        constexpr int NT = (TILE_E * 16 + 255) / 256;               // 8-byte pieces per thread
        constexpr int TQ = (FE_TILE_TAB_WORDS * 2 + 15) / 16;       // 16-byte pieces of the tables (<= 256)
        static_assert(TQ <= 256, "one table piece per thread");
        double tv[NT];
        const int tid = threadIdx.x;
        const uint4 tq = reinterpret_cast<const uint4*>(tb->t_m)[min(tid, TQ - 1)];
#pragma unroll
        for (int i = 0; i < NT; ++i) {                               // all loads in flight before the first is waited for
            const int idx = min(tid + 256 * i, TILE_E * 16 - 1);
            tv[i] = src[(idx >> 4) * 64 + (idx & 15)];
        }
        if (tid < TQ) reinterpret_cast<uint4*>(Tab)[tid] = tq;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int idx = tid + 256 * i;
            if (idx < TILE_E * 16) Tile[(idx >> 4) * FE_TILE_ROW + (idx & 15)] = tv[i];
        }
        __syncthreads();
    }

    double* Jq = Jl + (FROMQ ? 0 : q * ((K + 1) * N));      // (dense records only)
    uint32_t flags = 0;

    // ---- the inputs of the task-space signal FIRST: s_waitcnt counts loads in issue order, so whoever is requested last
    // waits for everything before it.  Requested ahead of the 8 rows of M and the 26 words of J per lane, the poses /
    // targets / gains are there one latency after the kernel starts and the ~480 instructions of the task error run while
    // the big loads are still arriving, instead of behind all of them.
    const bool has_tv = p.tvel != nullptr;
    const int dv = l >> 2, ang_id = l & 3;
    const int dd = dv < nd ? dv : nd - 1;
    TM ee_in[7];
    TIN tg_in[7], g_in[IRLOSC_GAIN_WORDS], tv_in[6];
    // Dense records: part 1 of the task signal may arrive as rows (Row16Extra::trows, osc_task_rows_dense_kernel).  Its ~405
    // instructions, run by sixteen lanes per instance here, cost this kernel 47 us per train on float64 records and 70 us on float32
    // records (where it is issue-bound); one lane per (instance, device) does them in a pass of ~30 us: 5.83 -> 5.99e8 and 7.19 ->
    // 7.59e8 steps/s, bit for bit the same torques (profiles/NOTES.md, round 5).  x.trows == nullptr (IRLOSC_TASK_PASS=0): computed here.
    bool pre = false;
    if constexpr (!FROMQ) pre = x.trows != nullptr;
    double trow_in = 0.0;
    {
        const TIN* __restrict__ tgp = p.tgt + ((size_t)bc * nd + dd) * 7;
        const TIN* __restrict__ gp = p.gains + (p.gains_per_instance ? (size_t)bc * nd * IRLOSC_GAIN_WORDS : 0) + dd * IRLOSC_GAIN_WORDS;
        if constexpr (FROMQ) {      // part 1 of the task signal was computed by the task pass: only the velocity gain is needed here
            g_in[1] = gp[1];
        } else if (pre) {
            g_in[1] = gp[1];
            trow_in = l < kr ? x.trows[(size_t)bc * 16 + l] : 0.0;
        } else {
            const TIN* __restrict__ eep = p.ee + ((size_t)bc * nd + dd) * 7;
#pragma unroll
            for (int i = 0; i < 7; ++i) ee_in[i] = eep[i];
#pragma unroll
            for (int i = 0; i < 7; ++i) tg_in[i] = tgp[i];
#pragma unroll
            for (int i = 0; i < IRLOSC_GAIN_WORDS; ++i) g_in[i] = gp[i];
        }
        const TIN* __restrict__ tvp = has_tv ? p.tvel + ((size_t)bc * nd + dd) * 6 : zeros;      // all six together: as a
#pragma unroll                                                                                      // short-circuit chain each
        for (int i = 0; i < 6; ++i) tv_in[i] = tvp[i];                                              // waited for the one before
    }

    // ---- prologue: first rows of M in flight, J (coalesced) into LDS, dq ------------------------------------------
    TM pm0[N], pm1[N];
    TM jl0[K], jl1[K];
    TM dq0_in, dq1_in;
    const bool use_g = (p.cfgflags & IRLOSC_USE_G) != 0;
    unsigned mo0 = 0, mo1 = 0;      // FROMQ: tile rows of the next row of M to be read (the table reads run one column ahead)
    constexpr int PFQ = 2;          // FROMQ: rows of M in flight ahead of the column being eliminated (LDS latency, not HBM's)
    if constexpr (FROMQ) {
        static_for<0, PFQ>([&](auto jc) {      // the recursion runs from the last column down
            constexpr int j = N - 1 - decltype(jc)::value;
            pm0[j] = tile_at(Tm[j * 32 + l]);
            if constexpr (!tree_row_slot1_zero<TOPO>(j)) pm1[j] = tile_at(Tm[j * 32 + 16 + l]);
        });
        constexpr int jn = N - 1 - PFQ;
        mo0 = Tm[jn * 32 + l];
        if constexpr (!tree_row_slot1_zero<TOPO>(jn)) mo1 = Tm[jn * 32 + 16 + l];
        dq0_in = x.qvel[(size_t)bc * N + l];
        dq1_in = v1 ? x.qvel[(size_t)bc * N + 16 + l] : 0.0;
    } else {
        if constexpr (TREE) {
            static_for<0, kTreeFetch<TOPO, IRLOSC_R16_TREE_BUDGET>.count[0]>([&](auto pc) {
                constexpr int j = N - 1 - decltype(pc)::value;
                pm0[j] = m0p[j * N];
                if constexpr (!tree_slot1_zero<TOPO>(j)) pm1[j] = m1p[j * N];
            });
        } else {
            static_for<0, PF>([&](auto jc) { constexpr int j = decltype(jc)::value; pm0[j] = m0p[j * N]; pm1[j] = m1p[j * N]; });
        }
        const TIN* __restrict__ Jb = p.J + (size_t)bc * (kr * N);
        const TIN* __restrict__ Jb1 = v1 ? Jb + 16 + l : zeros;
#pragma unroll
        for (int r = 0; r < K; ++r) {
            if constexpr (PAD) {      // rows k .. KMAX - 1: the page of zeros (the records hold k rows)
                jl0[r] = (r < kr ? Jb + r * N + l : zeros)[0];
                jl1[r] = (r < kr ? Jb1 + r * N : zeros)[0];
            } else { jl0[r] = Jb[r * N + l]; jl1[r] = Jb1[r * N]; }
        }
        dq0_in = p.dq[(size_t)bc * N + l];
        dq1_in = (v1 ? p.dq + (size_t)bc * N + 16 + l : zeros)[0];
    }
    Wl[q][l] = 0.0;
    lds_sync();

    // ---- task-space signal, part 1 (osc.py:101-118,70-99,160-168): quad d of the row = device d -------------------
    // Lane a of the quad evaluates ONE of the three Euler angles (the fp64 atan2 is the expensive part), the quad
    // broadcasts them, every lane of the quad finishes the gains, lane 0 parks the controlled rows in LDS.
    // Values that are needed once in the middle and once at the very end -- the null-space gain, the bias forces, the instance
    // id -- are fetched / recomputed WHERE they are used (cached lines, a handful of instructions) instead of riding through
    // the whole kernel in vector registers: `late()` makes the lane id opaque so that the compiler cannot share the early copy.
    auto late_lane = [&]() { int t2 = threadIdx.x; asm volatile("" : "+v"(t2)); return t2; };
    auto late_bc = [&](const int tid2) { return (int)blockIdx.x * (4 * NW) + (tid2 >> 4); };
    auto load_kvn = [&]() -> double {
        if (!(p.cfgflags & IRLOSC_NULLSPACE)) return 0.0;
        const int b2 = late_bc(late_lane());
        return (double)p.null_kv[p.gains_per_instance ? (b2 < p.B ? b2 : p.B - 1) : 0];
    };
    const bool has_wr = (p.cfgflags & IRLOSC_ADMITTANCE) && p.wrench != nullptr;
    const DevMeta dm = p.dev[dd];
    bool own_brB = false;
    if (FROMQ || pre) {
        // Part 1 arrives as k rows of the tile (osc_task_rows_fromq_kernel, one lane per robot, ran between the walk and this
        // kernel): lane l takes row l (rows >= k: the entry of zeros); the device lanes only leave the velocity gain and the
        // damping-branch verdict of their device (osc.py:173).  405 of the ~2 650 VALU instructions of a wave went into sixteen
        // lanes per robot each running the pipeline for one Euler angle.  (Computing the rows HERE, by wave 0 while the tile arrives,
        // was measured in round 5: the other three waves wait at the barrier for its ~450 dependent instructions -- profiles/NOTES.md.)
        bool all_nonzero = has_tv;
#pragma unroll
        for (int i = 0; i < 6; ++i) all_nonzero = all_nonzero & ((double)tv_in[i] != 0.0);
        own_brB = all_nonzero && dv < nd;            // np.all(target_vel) == 0 quirk, osc.py:173
        if constexpr (FROMQ) Wl[q][l] = tile_at(Te[l]);
        else Wl[q][l] = trow_in;               // (dense records with the task pass: rows >= k are zeros)
        if (ang_id == 0 && dv < nd) {
            Kvl[q][dv] = (double)g_in[1];
            Brl[q][dv] = all_nonzero ? 0 : 1;
        }
        if (own_brB) {
            flags |= IRLOSC_FLAG_VEL_BRANCH_B;
            if (dm.jidx0 + dm.rows > kr) flags |= IRLOSC_FLAG_BAD_JIDX;
        }
    } else {
        double ee[7], tg[7], g[IRLOSC_GAIN_WORDS];
#pragma unroll
        for (int i = 0; i < 7; ++i) { ee[i] = (double)ee_in[i]; tg[i] = (double)tg_in[i]; }
#pragma unroll
        for (int i = 0; i < IRLOSC_GAIN_WORDS; ++i) g[i] = (double)g_in[i];
        double e[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (dm.calc & 1u) { e[0] = ee[0] - tg[0]; e[1] = ee[1] - tg[1]; e[2] = ee[2] - tg[2]; }
        if (dm.calc & 2u) {
            // transforms3d calls of osc.py:115-117
            const TaskRot R = task_rot(ee, tg);
            double ay, ax;
            R.angle_args(ang_id, ay, ax);
            const double ang = atan2(ay, ax);
            e[3] = quad_bcast(ang, 0);
            e[4] = quad_bcast(ang, 1);
            e[5] = quad_bcast(ang, 2);
        }
        apply_gains6_fast(g, e);
        bool all_nonzero = has_tv;
#pragma unroll
        for (int i = 0; i < 6; ++i) all_nonzero = all_nonzero & ((double)tv_in[i] != 0.0);
        own_brB = all_nonzero && dv < nd;            // np.all(target_vel) == 0 quirk, osc.py:173
        if (ang_id == 0 && dv < nd) {
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if (dm.dofmask & (1u << i)) { Wl[q][dm.row0 + cnt] = e[i]; ++cnt; }
            Kvl[q][dv] = g[1];
            Brl[q][dv] = all_nonzero ? 0 : 1;
        }
        if (own_brB) {
            flags |= IRLOSC_FLAG_VEL_BRANCH_B;
            if (dm.jidx0 + dm.rows > kr) flags |= IRLOSC_FLAG_BAD_JIDX;
        }
    }
    IRLOSC_TS(1);
    // J into LDS (its loads have been in flight since the top of the kernel); FROMQ: J already sits in the tile
    if constexpr (!FROMQ) {
#pragma unroll
        for (int r = 0; r < K; ++r) Jq[r * N + l] = (double)jl0[r];
        Jq[K * N + l] = 0.0;
        if (v1) {
#pragma unroll
            for (int r = 0; r < K; ++r) Jq[r * N + 16 + l] = (double)jl1[r];
            Jq[K * N + 16 + l] = 0.0;
        }
    }
    const double dq0 = (double)dq0_in, dq1 = (double)dq1_in;
    lds_sync();
    __builtin_amdgcn_sched_barrier(0);

    IRLOSC_TS(2);
    // ---- main loop: Cholesky of M, Y = L^-1 J^T (as T), M dq, J dq ------------------------------------------------
    double T[N];
    double mdq0 = 0.0, mdq1 = 0.0, dx = 0.0;
    unsigned long long npd_mask = 0;      // lanes that saw a non-positive pivot of M (scalar registers: see pinned_ballot)
    const double* trow = Jq + (l < K ? l : K) * N;
    // column j of J in the right-hand-side layout: J[r][j] for lane r (rows K .. 15 of the tile table name the entry of zeros)
    auto jcol = [&](const int j) -> double {
        if constexpr (FROMQ) return tile_at(Tj[l * 32 + j]);
        else return trow[j];
    };
    if constexpr (TREE) {
        // M = L^T L, columns N - 1 .. 0; R0[c] = L[c][l], R1[c] = L[c][16 + l] (the lane's two COLUMNS of L)
        using TI = FeTopo<TOPO>;
        static_assert(TOPO::NJ == N, "tree shape and kernel shape");
        double R0[N], R1[N];
        double tnext = jcol(N - 1);
        static_for_down<0, N>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int sj = j >> 4, gj = j & 15;
            constexpr int SZ = tree_subtree_size<TOPO>(j);          // hinges j .. j + SZ - 1 are the subtree of j
            static_assert(tree_subtree_contiguous<TOPO>(j), "depth-first numbering: a subtree is a run of indices");
            constexpr bool EEJ = tree_moves_ee<TOPO>(j);            // otherwise column j of J and row j of Y are zero
            constexpr bool S1Z = tree_slot1_zero<TOPO>(j);          // no slot-1 half: m1 = 0, nothing loaded for it
            if constexpr (FROMQ) {
                if constexpr (j - PFQ >= 0) {
                    pm0[j - PFQ] = tile_at(mo0);
                    if constexpr (!tree_slot1_zero<TOPO>(j - PFQ)) pm1[j - PFQ] = tile_at(mo1);
                    if constexpr (j - PFQ - 1 >= 0) {
                        mo0 = Tm[(j - PFQ - 1) * 32 + l];
                        if constexpr (!tree_slot1_zero<TOPO>(j - PFQ - 1)) mo1 = Tm[(j - PFQ - 1) * 32 + 16 + l];
                    }
                }
            }
            double m0 = (double)pm0[j], m1 = 0.0;
            if constexpr (!S1Z) m1 = (double)pm1[j];
            if constexpr (!FROMQ) {      // dense records: the rows the prefetch plan releases now that row j's registers are free
                constexpr int e = N - j;                            // step N - 1 - j, entry e = step + 1
                static_for<0, kTreeFetch<TOPO, IRLOSC_R16_TREE_BUDGET>.count[e]>([&](auto pc) {
                    constexpr int jn = N - 1 - (kTreeFetch<TOPO, IRLOSC_R16_TREE_BUDGET>.first[e] + decltype(pc)::value);
                    pm0[jn] = m0p[jn * N];
                    if constexpr (!tree_slot1_zero<TOPO>(jn)) pm1[jn] = m1p[jn * N];
                });
            }
            double tj = EEJ ? tnext : 0.0;
            if constexpr (j > 0) { if constexpr (tree_moves_ee<TOPO>(j - 1)) tnext = jcol(j - 1); }
            const double dqs = sj ? dq1 : dq0;
            __builtin_amdgcn_sched_barrier(0);
            fmac_bc_nop<gj>(mdq0, dqs, m0);
            if constexpr (!S1Z) fmac_bc<gj>(mdq1, dqs, m1);
            if constexpr (EEJ) fmac_bc<gj>(dx, dqs, tj);
            tree_chains<TOPO, j, j + 1, SZ - 1>(m0, m1, tj, R0, R1, T);
            double d = bc_nop<gj>(sj ? m1 : m0);
            npd_mask |= pinned_ballot(!(d > 0.0));      // also catches NaN
            d = fmax(d, 1e-300);
            const double dinv = rsq_refined(d);
            R0[j] = m0 * dinv;
            if constexpr (j >= 16) R1[j] = m1 * dinv;
            T[j] = tj * dinv;
            __builtin_amdgcn_sched_barrier(0);
        });
    } else {
    double L0[15], L1[24];
    double tnext = trow[0];
    static_for<0, N>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int sj = j >> 4, gj = j & 15;
        if constexpr (j + PF < N) { pm0[j + PF] = m0p[(j + PF) * N]; pm1[j + PF] = m1p[(j + PF) * N]; }
        double m0 = (double)pm0[j], m1 = (double)pm1[j];
        double tj = tnext;
        if constexpr (j + 1 < N) tnext = trow[j + 1];
        const double dqs = sj ? dq1 : dq0;
        __builtin_amdgcn_sched_barrier(0);
        fmac_bc_nop<gj>(mdq0, dqs, m0);
        fmac_bc<gj>(mdq1, dqs, m1);
        fmac_bc<gj>(dx, dqs, tj);
        // the column terms, one asm statement per chunk of them (osc_row16_asm.hpp: no compiler padding inside)
        if constexpr (j > 0 && j < 16) fmac3_chain<gj, 0, j>(m0, m1, tj, L0, L1, T);
        else if constexpr (j >= 16) fmac2_chain<gj, 0, j>(m1, tj, L1, T);
        double d = bc_nop<gj>(sj ? m1 : m0);
        npd_mask |= pinned_ballot(!(d > 0.0));      // also catches NaN
        d = fmax(d, 1e-300);
        const double dinv = rsq_refined(d);
        if constexpr (j < 15) L0[j] = m0 * dinv;
        if constexpr (j < 24) L1[j] = m1 * dinv;
        T[j] = tj * dinv;
        __builtin_amdgcn_sched_barrier(0);
    });
    }

    flags |= ((npd_mask >> lane) & 1ull) ? IRLOSC_FLAG_M_NOT_PD : 0u;
    IRLOSC_TS(3);
    // The admittance wrench (osc.py:184-185) is consumed behind A = Y^T Y: requested HERE, where the registers of the factor have
    // just come free, its round trip to HBM hides behind those K x N products.  (Round 4 loaded it where it is added: the k12 +
    // admittance kernel issued in 56 % of its cycles against 73 % for k13 without the term -- profiles/r05b_row16_f64_k12_admit_pmc_sq.txt.)
    TIN wr_in[6];
    if (has_wr) {
        const TIN* __restrict__ wp = p.wrench + ((size_t)bc * nd + dd) * 6;
#pragma unroll
        for (int i = 0; i < 6; ++i) wr_in[i] = wp[i];
    }
    // ---- A = Y^T Y: lane c ends up with A[r][c], r = 0..K-1 -------------------------------------------------------
    double A[K];
#pragma unroll
    for (int r = 0; r < K; ++r) A[r] = 0.0;
    static_for<0, N>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (tree_row_of_y<TOPO>(i)) {               // tree: the rows of Y under no end effector are zero
            static_for<0, K>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                // (The task rows of one arm only see that arm's hinges and the stand: with the rows grouped by arm, 85 of these 169
                // products are structurally non-zero.  Measured as an upper bound with the bench's row order hard-wired: +1.2 % --
                // not worth a row permutation and a second instantiation per shape.)
                if constexpr (i == 0 && r == 0) fmac_bc_nop<r>(A[r], T[i], T[i]);
                else fmac_bc<r>(A[r], T[i], T[i]);
            });
        }
    });
    __builtin_amdgcn_sched_barrier(0);

    IRLOSC_TS(4);
    // ---- task-space signal, part 2: target-velocity branch B and the admittance wrench (osc.py:173-185) -----------
    Dxl[q][l] = dx;
    lds_sync();
    if ((own_brB || has_wr) && ang_id == 0 && dv < nd) {
        const double kv = Kvl[q][dv];
        const TIN* __restrict__ gp = p.gains + (p.gains_per_instance ? (size_t)bc * nd * IRLOSC_GAIN_WORDS : 0) + dd * IRLOSC_GAIN_WORDS;
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (dm.dofmask & (1u << i)) {
                double v = Wl[q][dm.row0 + cnt];
                if (own_brB) {
                    const int row = dm.jidx0 + cnt;
                    const double dxv = row < kr ? Dxl[q][row] : 0.0;
                    const double damp = i < 3 ? (double)gp[6 + i] : 1.0;
                    v += kv * (dxv - (double)p.tvel[((size_t)bc * nd + dd) * 6 + i]) * damp;
                }
                if (has_wr) v += (double)wr_in[i];
                Wl[q][dm.row0 + cnt] = v;
                ++cnt;
            }
        }
    }
    lds_sync();
    // w = u_task_all [+ ext_f] - kvn * dx  (null-space term folded in: osc_generic.hpp header)
    double w = Wl[q][l] - load_kvn() * dx;
    // M dq is next needed for the torques at the very end: it waits in the two LDS rows that are free from here on (every
    // lane its own slot; all cross-lane reads of Wl / Dxl are behind the lds_sync above)
    Dxl[q][l] = mdq0;
    Wl[q][l] = mdq1;

    // ---- k x k: A = L~ D L~^T, column c (= row c) of everything in lane c -------------------------------------------
    double nA2 = 0.0;
#pragma unroll
    for (int r = 0; r < K; ++r) nA2 = fma(A[r], A[r], nA2);
    nA2 = row_sum(nA2);
    double Ac[K];           // A itself, for the eigen path (the factorisation runs in place)
#pragma unroll
    for (int r = 0; r < K; ++r) Ac[r] = A[r];
    double F[K], G[K];      // rows / columns of L~ (see ldl16)
    double invd_own = 0.0;  // 1 / d_c in lane c
    double detA = 1.0;
    bool pdA = true;
    // PAD: task rows that NO joint can move (row r of J identically zero, e.g. a base device asked for translations, or any k > 13
    // on this robot: 13 columns of J can be non-zero) are exact null directions of A: the reference's det is 0, its pinv drops them
    // (osc.py:52-55).  They are taken out here like the padding rows, instead of being found one by one by the eigen stage (three at most).
    uint32_t zrow = 0u;
    bool zr = false;
    if constexpr (PAD) {
        double dg = 1.0;
#pragma unroll
        for (int r = 0; r < K; ++r) dg = (l == r) ? A[r] : dg;
        zr = l < kr && dg == 0.0;                       // A[r][r] = |row r of Y|^2: zero exactly when row r of J is zero
        zrow = (uint32_t)(__ballot(zr) >> (q * 16)) & 0xffffu;
        w = zr ? 0.0 : w;
    }
    ldl16<K, PAD>(A, l, 0.0, F, G, invd_own, pdA, detA, kr, zrow);
    // W = L~^-1, row c in lane c: X[m] = W[c][m]
    double X[K];
#pragma unroll
    for (int m = 0; m < K; ++m) X[m] = (l == m) ? 1.0 : 0.0;
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, K - 1>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        static_for<0, j + 1>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            if constexpr (j < 3 || m == 0) fmac_bc_n_nop<j>(X[m], X[m], F[j]);
            else fmac_bc_n<j>(X[m], X[m], F[j]);
        });
    });
    __builtin_amdgcn_sched_barrier(0);
    double trA = 0.0;
#pragma unroll
    for (int m = 0; m < K; ++m) trA = fma(X[m], X[m], trA);
    if constexpr (PAD) trA = (l < kr && !zr) ? trA : 0.0;      // (the padded block is the identity: not part of the trace)
    trA = row_sum(trA * invd_own);                     // trace(A^-1) >= 1 / lambda_min
    const bool small_det = !pdA || !(fabs(detA) >= 1e-4) || zrow != 0u;      // a non-positive pivot: A is singular to working precision
    const double cond_bound = sqrt(nA2) * trA;         // >= cond_2(A) for SPD A
    const bool plain = pdA && t_finite(cond_bound) && (!small_det || cond_bound < 0.99e5);
    flags |= small_det ? IRLOSC_FLAG_PINV_BRANCH : 0u;
    flags |= plain ? 0u : IRLOSC_FLAG_EIGEN_PATH;
    flags |= zrow != 0u ? IRLOSC_FLAG_TRUNCATED : 0u;       // (the exact zero eigenvalues are under any cut)
    // t = A^-1 w = L~^-T D^-1 (W w)
    double z = 0.0;
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, K>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        if constexpr (m == 0) fmac_bc_nop<m>(z, w, X[m]);
        else fmac_bc<m>(z, w, X[m]);
    });
    double t = z * invd_own;
    __builtin_amdgcn_sched_barrier(0);
    static_for_down<1, K>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        fmac_bc_n_nop<i>(t, t, G[i]);
    });
    IRLOSC_TS(5);
    // Instances that are not certifiably on the reference's inverse branch: truncated pseudo-inverse (osc.py:55).
    // Wave-uniform branch: the whole wave runs it (DPP sources must be active lanes), the others keep their t.
    // (Handing the flagged instances to a pass of their own, four to a wave, was built and measured in round 3: the main
    // kernel drops from 1 212 to 1 045 us per train of 8 and the pass costs 140 us -- zero-sum.  eigen16 is chains of
    // dependent broadcast-FMAs that issue at a fraction of the main loop's rate; here they hide behind the wave-mate's
    // main loop, in a pass of their own two such chains share a SIMD.  profiles/r03c_eigen_handover_experiment_*.  Round 5 repeated
    // it for the fused path, whose blocks of four waves wait for their slowest wave: OSC kernel 821 -> 637 us, pass 176 us at three
    // waves per SIMD -- the stage costs the same SIMD time per flagged robot wherever it runs.  profiles/NOTES.md.)
    bool giveup = false;
    if (__any(!plain)) {
        double t2 = 0.0;
        uint32_t f2 = 0;
        eigen16<K, PAD>(Ac, F, G, invd_own, pdA, nA2, trA, w, l, !plain, t2, f2, giveup, kr, zrow);
        t = plain ? t : t2;
        flags |= plain ? 0u : f2;
    }

    IRLOSC_TS(6);
    // ---- joint torques of the own rows: u = u0 + bias - kvn * Mdq - J^T t (osc.py:174,184-200) --------------------
    double jt0 = 0.0, jt1 = 0.0;
    const int lane2 = late_lane(), l2 = lane2 & 15;
    const int b2 = late_bc(lane2);
    const bool live2 = b2 < p.B;
    const int bc2 = live2 ? b2 : p.B - 1;
    TM bias0_in, bias1_in;                 // requested here, consumed behind the J^T t products
    // (FROMQ: the tile addresses of this phase are rebuilt from the late copy of the thread id, like everything else down here)
    const unsigned rob2 = (unsigned)(lane2 >> 4) * 8u;
    auto tile2 = [&](const unsigned row_off) -> double {
        return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(Tile) + (row_off + rob2));
    };
    if constexpr (FROMQ) {
        bias0_in = use_g ? tile2(Tb[l2]) : 0.0;
        bias1_in = use_g ? tile2(Tb[16 + l2]) : 0.0;      // joints >= N: the entry of zeros
    } else {
        bias0_in = (use_g ? p.bias + (size_t)bc2 * N + l2 : zeros)[0];
        bias1_in = (use_g && l2 < N1 ? p.bias + (size_t)bc2 * N + 16 + l2 : zeros)[0];
    }
    const double kvn = load_kvn();
    // (the LDS rows of this wave, addressed from the late thread id: the early base addresses would ride through the eigen stage)
    const int wv2 = FROMQ ? (lane2 >> 6) : 0, q2 = (lane2 >> 4) & 3;
    mdq0 = Dxl_[wv2][q2][l2];
    mdq1 = Wl_[wv2][q2][l2];
    {
        double jr0[K], jr1[K];
#pragma unroll
        for (int r = 0; r < K; ++r) {
            if constexpr (FROMQ) { jr0[r] = tile2(Tj[r * 32 + l2]); jr1[r] = tile2(Tj[r * 32 + 16 + l2]); }      // joints >= N: zeros
            else { jr0[r] = Jq[r * N + l]; jr1[r] = Jq[r * N + 16 + l]; }   // padding lanes: junk, never stored
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, K>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            if constexpr (r == 0) fmac_bc_nop<r>(jt0, t, jr0[r]);
            else fmac_bc<r>(jt0, t, jr0[r]);
            fmac_bc<r>(jt1, t, jr1[r]);
        });
    }
    double u0 = 0.0, u1 = 0.0;
#pragma unroll
    for (int d2 = 0; d2 < NDEV; ++d2) {       // branch A damping, osc.py:174 (assignment, device order)
        if (PAD && d2 >= nd) break;
        const bool brA = Brl_[wv2][q2][d2] != 0;
        const double kvd = Kvl_[wv2][q2][d2];
        const uint32_t jm = p.dev[d2].joint_mask;
        if (brA && ((jm >> l) & 1u)) u0 = -kvd * mdq0;
        if (brA && ((jm >> ((16 + l) & 31)) & 1u)) u1 = -kvd * mdq1;
    }
    u0 += (double)bias0_in;          // zeros unless IRLOSC_USE_G
    u1 += (double)bias1_in;
    u0 -= kvn * mdq0;
    u1 -= kvn * mdq1;
    u0 -= jt0;                       // the task term last
    u1 -= jt1;
    const bool bad = !t_finite(u0) || (v1 && !t_finite(u1));
    flags |= bad ? IRLOSC_FLAG_NONFINITE : 0u;
    // flags of the instance = OR over its 16 lanes
#pragma unroll
    for (int bit = 0; bit < 7; ++bit) {
        const unsigned long long m = __ballot((flags >> bit) & 1u);
        if ((m >> (q * 16)) & 0xffffull) flags |= 1u << bit;
    }
    if (live2) {
        p.u[(size_t)b2 * N + l2] = (TIN)u0;
        if (l2 < N1) p.u[(size_t)b2 * N + 16 + l2] = (TIN)u1;
        if (l2 == 0) {
            p.flags[b2] = flags;
            if (giveup) x.worklist[atomicAdd(x.workcount, 1)] = b2;
        }
    }
    IRLOSC_TS(7);
    if (x.span && lane == 0) {       // first wave's start / last wave's end of the train, untraced (irlosc_time_trains)
        unsigned long long* sp = x.span + 2 * (blockIdx.x & (R16_SPAN_SLOTS - 1));
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        atomicMin(sp, rt0);
        atomicMax(sp + 1, rt1);
        if (blockIdx.x == 0 && blockIdx.y == 0 && wv == 0) {      // one sample wave per train: shader cycles against wall clock
            x.span[2 * R16_SPAN_SLOTS] = (unsigned long long)__builtin_readcyclecounter() - cyc0;
            x.span[2 * R16_SPAN_SLOTS + 1] = rt1 - rt0;
        }
    }
    if (p.dbg && lane == 0) {
        const size_t wid = (size_t)blockIdx.x * NW + wv;
#pragma unroll
        for (int i = 0; i < 8; ++i) p.dbg[wid * 10 + i] = ts[i];      // (the last step of a train wins)
        p.dbg[wid * 10 + 8] = rt0;
        p.dbg[wid * 10 + 9] = __builtin_amdgcn_s_memrealtime();
    }
#undef IRLOSC_TS
}

// The generic kernel over a worklist: instance ids list[0..*count); zeroes *reset for the step after.
// T = arithmetic type, S = storage type of the records (S = float, T = double on the mixed path).
template <typename T, typename S>
__global__ __launch_bounds__(64) void osc_generic_worklist_kernel(const Row16Train<S> tr, int32_t* __restrict__ reset) {   // reset: optional, R16_TRAIN counters to zero
    extern __shared__ __align__(16) unsigned char smem_raw_w[];
    T* smem = reinterpret_cast<T*>(smem_raw_w);
    const KParams<S>& p = tr.p[blockIdx.y];
    const int32_t* __restrict__ list = tr.x[blockIdx.y].worklist;
    const int n = *tr.x[blockIdx.y].workcount;
    // (The give-up counters of a train are zeroed by the host with a memset in front of its main kernel -- ALL R16_TRAIN of
    // them, whatever the train's length: a shorter train must not inherit what a longer one counted.)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < R16_TRAIN && reset) reset[threadIdx.x] = 0;
    for (int it = blockIdx.x; it < n; it += gridDim.x) generic_instance<T>(p, list[it], smem);
}

// FROMQ: part 1 of the task-space signal (calc_error, velocity limit, gains, stiffness: osc.py:101-118,70-99,160-168) as a PASS OF ITS
// OWN between the walk and the OSC kernel -- ONE LANE PER ROBOT, block x = walk wave x (its 64 robots), blockIdx.y = step.  Reads
// the end-effector poses the walk parked (coalesced: [entry][64 robots]), the targets and gains, and leaves the k gained error rows
// as k more entries of the exchange block (FeTopo::task_index), which the OSC kernel's tile picks up like everything else.  Same
// formulas as the in-kernel form of the dense-record path (task_rot, apply_gains6_fast).  ~450 instructions per wave of 64
// (robot, device) pairs against ~400 per wave of FOUR robots in the OSC kernel.
// PAD: one instantiation for every layout (NDEV = IRLOSC_MAX_DEV; the block has 64 x p.ndev threads).
template <int K, int NDEV, typename TIN, class TOPO, bool PAD = false>
__global__ __launch_bounds__(64 * NDEV) void osc_task_rows_fromq_kernel(const Row16Train<TIN> tr) {
    using namespace r16;
    const KParams<TIN>& p = tr.p[blockIdx.y];
    const Row16Extra& x = tr.x[blockIdx.y];
    const int nd = PAD ? p.ndev : NDEV;
    // block = the 64 robots of walk wave blockIdx.x x NDEV waves: wave d computes the rows of target device d.  (The pass moves 230 MB
    // per train of 8 -- poses in, targets in, rows out -- and takes 60 us either way, one wave per robot or per (robot, device).)
    const int lane = threadIdx.x & 63;
    const int d = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x * 64 + lane;
    const int bc = b < p.B ? b : p.B - 1;        // idle lanes of a ragged last block: the walk left the last robot's data in their columns
    const FeCompactTables* __restrict__ tb = x.tables;
    constexpr int BLK_E = FeTopo<TOPO>::n_compact();      // (through a constexpr VARIABLE: called inside a runtime expression, the
                                                          // constexpr function is evaluated at run time -- loops over the tree, 2.4 ms)
    double* __restrict__ col = const_cast<double*>(x.side) + (size_t)blockIdx.x * BLK_E * 64 + lane;
    // The targets of the block's 64 robots are 64 x NDEV x 7 consecutive words: wave loads, all in flight together, transposed
    // through LDS (read straight per lane -- 168-byte strides, 64 lines per load instruction -- the texture addresser handles a
    // line per cycle).  Wave d brings in the d-th third of them.
    const int TW = nd * 7;                         // NDEV = 3: odd, conflict-free reads with the robot as the slow index
    __shared__ double tgs[64 * NDEV * 7];
    {
        const size_t g0 = (size_t)blockIdx.x * 64 * TW + (size_t)d * 64 * 7, glast = (size_t)p.B * TW - 1;
        TIN tv[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) { const size_t g = g0 + lane + 64 * i; tv[i] = p.tgt[g < glast ? g : glast]; }
#pragma unroll
        for (int i = 0; i < 7; ++i) tgs[d * 64 * 7 + lane + 64 * i] = (double)tv[i];
    }
    const unsigned e0 = tb->e0;                    // (a value, not `tb->e0` at every store: the stores might alias the table for all the compiler knows)
    uint16_t eet[7];                               // entry indices of the pose: requested together, ahead of the pose
#pragma unroll
    for (int i = 0; i < 7; ++i) eet[i] = tb->eetab[d][i];
    const DevMeta dm = p.dev[d];
    const TIN* __restrict__ gp = p.gains + (p.gains_per_instance ? (size_t)bc * nd * IRLOSC_GAIN_WORDS : 0) + d * IRLOSC_GAIN_WORDS;
    double ee[7], tg[7], g[IRLOSC_GAIN_WORDS];
#pragma unroll
    for (int i = 0; i < 7; ++i) ee[i] = col[(size_t)eet[i] * 64];
#pragma unroll
    for (int i = 0; i < IRLOSC_GAIN_WORDS; ++i) g[i] = (double)gp[i];
    __syncthreads();                               // the targets of all devices are in LDS
#pragma unroll
    for (int i = 0; i < 7; ++i) tg[i] = tgs[lane * TW + d * 7 + i];
    double e[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (dm.calc & 1u) { e[0] = ee[0] - tg[0]; e[1] = ee[1] - tg[1]; e[2] = ee[2] - tg[2]; }
    if (dm.calc & 2u) {
        const TaskRot R = task_rot(ee, tg);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double ay, ax;
            R.angle_args(a, ay, ax);
            e[3 + a] = atan2(ay, ax);
        }
    }
    apply_gains6_fast(g, e);
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (dm.dofmask & (1u << i)) { col[(size_t)(e0 + dm.row0 + cnt) * 64] = e[i]; ++cnt; }
}

// Dense records: part 1 of the task-space signal (calc_error, velocity limit, gains, stiffness: osc.py:101-118,
// 70-99,160-168) as a pass ahead of the row16 kernel -- ONE LANE PER (INSTANCE, DEVICE): block x = instances 64 x .. 64 x + 63, wave d =
// target device d (the block has 64 x ndev threads), blockIdx.y = step.  Reads pose, target and gains of its pair, leaves the k gained
// error rows of the block's instances in Row16Extra::trows ([B][16] doubles; rows >= k: zeros), transposed through LDS so that the
// 8 KB go out as whole lines.  Same formulas, in the same order, as the in-kernel form (task_rot, atan2, apply_gains6_fast): the
// row16 kernel's results do not depend on which of the two ran.
template <typename TIN>
__global__ __launch_bounds__(64 * IRLOSC_MAX_DEV) void osc_task_rows_dense_kernel(const Row16Train<TIN> tr) {
    using namespace r16;
    const KParams<TIN>& p = tr.p[blockIdx.y];
    const Row16Extra& x = tr.x[blockIdx.y];
    const int nd = p.ndev;
    const int lane = threadIdx.x & 63;
    const int d = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x * 64 + lane;
    const int bc = b < p.B ? b : p.B - 1;
    // The poses and targets of the block's 64 instances are 64 x ndev x 7 consecutive words each: wave loads, all in flight together,
    // transposed through LDS.  (Read straight per lane -- 7 x ndev-word strides, 64 lines per load instruction -- the texture addresser
    // handles a line per cycle: 14 such loads per wave made this pass 63 us per train where its bytes take 45.)
    constexpr int WMAX = 64 * IRLOSC_MAX_DEV * 7;
    constexpr size_t SB = 2 * WMAX * sizeof(TIN) > 64 * 17 * sizeof(double) ? 2 * WMAX * sizeof(TIN) : 64 * 17 * sizeof(double);
    __shared__ __align__(16) unsigned char smem_t[SB];              // poses + targets; afterwards the rows, [instance][16 + 1 pad]
    TIN* const s_ee = reinterpret_cast<TIN*>(smem_t);
    TIN* const s_tg = s_ee + WMAX;
    double (*rows)[17] = reinterpret_cast<double (*)[17]>(smem_t);
    {
        const int T = 64 * nd;                                       // threads of the block = words per round
        const size_t g0 = (size_t)blockIdx.x * 64 * nd * 7, glast = (size_t)p.B * nd * 7 - 1;
        TIN ve[7], vt[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) { const size_t g = g0 + threadIdx.x + (size_t)T * i; ve[i] = p.ee[g < glast ? g : glast]; }
#pragma unroll
        for (int i = 0; i < 7; ++i) { const size_t g = g0 + threadIdx.x + (size_t)T * i; vt[i] = p.tgt[g < glast ? g : glast]; }
#pragma unroll
        for (int i = 0; i < 7; ++i) { s_ee[threadIdx.x + T * i] = ve[i]; s_tg[threadIdx.x + T * i] = vt[i]; }
    }
    const DevMeta dm = p.dev[d];
    const TIN* __restrict__ gp = p.gains + (p.gains_per_instance ? (size_t)bc * nd * IRLOSC_GAIN_WORDS : 0) + d * IRLOSC_GAIN_WORDS;
    double g[IRLOSC_GAIN_WORDS];
#pragma unroll
    for (int i = 0; i < IRLOSC_GAIN_WORDS; ++i) g[i] = (double)gp[i];
    __syncthreads();                               // poses and targets are in LDS
    double ee[7], tg[7];
    // (a ragged last block: the lanes past the batch read the words the clamped loads left -- finite, never written out)
#pragma unroll
    for (int i = 0; i < 7; ++i) { ee[i] = (double)s_ee[(lane * nd + d) * 7 + i]; tg[i] = (double)s_tg[(lane * nd + d) * 7 + i]; }
    double e[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (dm.calc & 1u) { e[0] = ee[0] - tg[0]; e[1] = ee[1] - tg[1]; e[2] = ee[2] - tg[2]; }
    if (dm.calc & 2u) {
        const TaskRot R = task_rot(ee, tg);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double ay, ax;
            R.angle_args(a, ay, ax);
            e[3 + a] = atan2(ay, ax);
        }
    }
    apply_gains6_fast(g, e);
    __syncthreads();                               // every wave has its poses and targets in registers: the rows take their place
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (dm.dofmask & (1u << i)) { rows[lane][dm.row0 + cnt] = e[i]; ++cnt; }
    __syncthreads();
    double* __restrict__ out = const_cast<double*>(x.trows) + (size_t)blockIdx.x * (64 * 16);
    const size_t last = (size_t)p.B * 16;
    const int k = p.k;                             // (every row < k belongs to one device and has been written; the others are zeros)
    for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x)
        if ((size_t)blockIdx.x * (64 * 16) + i < last) out[i] = (i & 15) < k ? rows[i >> 4][i & 15] : 0.0;
}

// Shapes with an instantiation of their own (tuned: register budget, prefetch depth) ...
inline bool row16_kernel_exact(int n, int k, int ndev) {
    const char* e = getenv("IRLOSC_FORCE_PAD");      // A/B aid, read by irlosc_create only: the padded variant on a shape that has an instantiation
    if (e && e[0] == '1') return false;
    return n == 25 && ((k == 13 && ndev == 3) || (k == 12 && ndev == 2) || (k == 7 && ndev == 3) || (k == 6 && ndev == 2));
}
// ... and the KMAX-padded variants that take every other n = 25 layout: the smallest tier that holds k
constexpr int R16_PAD_TIERS[] = {4, 7, 10, 13, 16};
inline int row16_pad_tier(int k) {
    for (int t : R16_PAD_TIERS) if (k <= t) return t;
    return 0;
}
inline bool row16_kernel_supports(int dtype, int n, int k, int ndev) {
    (void)dtype;     // fp64 records, or fp32 records with fp64 arithmetic (mixed path)
    return n == 25 && k >= 1 && k <= IRLOSC_MAX_K && ndev >= 1 && ndev <= IRLOSC_MAX_DEV;
}

}  // namespace irlosc
