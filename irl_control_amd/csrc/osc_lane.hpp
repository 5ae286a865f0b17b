// The OSC step of the fused path in LANE-PER-ROBOT form (round 6): one lane = one robot, 64 robots per wavefront, the tree and the
// task-row structure compiled in.  It replaces the row16 FROMQ kernel behind the lane-per-robot walk (osc_frontend_lane.hpp) for the
// layouts it has an instantiation for; every other layout keeps the FROMQ kernel.
//
// Why.  The row16 form gives a robot 16 lanes and runs a 25-row / 13-row problem on them at ~25 % lane efficiency: 2 300 VALU
// instructions per wave of FOUR robots = 36 800 per 64 robots, three times the issue slots of the whole rigid-body walk
// (profiles/NOTES.md, round 5).  The exchange buffer the walk leaves is already laid out [entry][64 robots] --
// the native layout for one lane per robot: every operand is one coalesced 512-byte load, every matrix entry is a register with a
// compile-time index, and the sparsity of the tree AND of the task rows can be used entry by entry:
//
//   M = L^T L from the leaves up (hinges NJ-1 .. 0, no fill-in: osc_row16.hpp, TOPO) in right-looking "accumulated update" form:
//       row j of M is read when hinge j is eliminated, m_ji - Delta[j][i], Delta[a][b] += L[c][a] L[c][b] for the hinges c below
//       both -- a Delta is live from the first hinge under both until its own row is eliminated: <= 36 of the 180 at any time
//   Y = L^-T J^T rides along the same way (DJ): row r of J only has entries at the hinges that move ITS end effector
//       (device.py:125-133), so an arm's six rows see seven hinges, not twenty-five
//   A = Y^T Y = blockdiag(arm blocks) + y0 y0^T: rows of different end effectors only meet at their common hinges (the stand)
//   k x k: L~ D L~^T, trace(A^-1) by columns of L~^-1, the certificate and the plain solve of osc_row16.hpp (osc.py:51-55) --
//       ~1 500 lane-instructions where the row16 form spends ~700 per FOUR robots
//   u = u0 + bias - kvn M dq - J^T t (osc.py:148-200), transposed through LDS and stored row-major, coalesced.
//
// The task rows are kept in a CANONICAL order (grouped by end-effector body, padded to the instantiation's rows per body), not in
// targets order: A, w and t are permuted consistently (t = Mx w does not depend on the order of the rows), J^T t is a sum over
// rows.  Padded rows and rows no joint can move (A[r][r] == 0 exactly) are taken out of the factorisation like the KMAX-padded
// row16 kernels do (pivot 1, not part of det / trace; PINV and TRUNCATED set for an exact zero row as the reference's det = 0 /
// pinv imply).
//
// Robots whose solve is not certifiably the reference's inverse branch (15 % of physical states) are NOT finished here: the lane
// writes A, w and the Jacobian columns into a compact record (slot from one atomic per wave; records transposed in groups of 64) and
// osc_lane_eigen_kernel -- again one lane per robot, 64 flagged robots per wave, r16::eigen16's algorithm on per-lane arrays -- computes
// t = pinv(A) w (osc.py:55) and subtracts J^T t from the torques this kernel left without the task term.  What that stage gives up on
// goes to the generic kernel as before.
#pragma once
#include "osc_generic.hpp"
#include "osc_row16.hpp"
#include "osc_lane_types.hpp"

namespace irlosc {
namespace lane {

// Rows per end-effector candidate of the tree (bodies with TOPO::ee_cand, in body order) an instantiation holds
template <int... R>
struct Shape {
    static constexpr int NC = sizeof...(R);
    static constexpr int rows[NC] = {R...};
    static constexpr int K = (R + ...);
};

template <class TOPO, class SH>
struct LT {
    using TI = FeTopo<TOPO>;
    static constexpr int NJ = TOPO::NJ;
    static constexpr int K = SH::K;
    static constexpr int n_cand() { int n = 0; for (int b = 0; b < TOPO::NB; ++b) n += TOPO::ee_cand[b] ? 1 : 0; return n; }
    static constexpr int cand_body(int c) {
        int n = 0;
        for (int b = 0; b < TOPO::NB; ++b) if (TOPO::ee_cand[b]) { if (n == c) return b; ++n; }
        return -1;
    }
    static constexpr int row_cand(int r) { int c = 0; while (r >= SH::rows[c]) { r -= SH::rows[c]; ++c; } return c; }
    static constexpr int row_body(int r) { return cand_body(row_cand(r)); }
    static constexpr bool row_moved(int r, int j) { return TI::moves(j, row_body(r)); }                 // J[r][j] may be non-zero
    static constexpr bool hinge_ee(int j) { for (int r = 0; r < K; ++r) if (row_moved(r, j)) return true; return false; }
    static constexpr int ee_rank(int j) { int n = 0; for (int i = 0; i < j; ++i) n += hinge_ee(i) ? 1 : 0; return n; }      // position among the EE hinges
    static constexpr int n_ee() { return ee_rank(NJ); }
    static constexpr int jentry0(int r, int j) { return TI::ee_index(row_body(r)) + 7 + 6 * TI::anc_rank(j, row_body(r)); }
    static constexpr int subtree_last(int j) { int l = j; for (int c = j; c < NJ; ++c) if (TI::above(j, c)) l = c; return l; }
    static constexpr bool has_below(int j) { return subtree_last(j) > j; }
    // the deepest hinge (largest index = first in the descending recursion) that moves row r's body, resp. both bodies
    static constexpr int deepest(int r) { int d = -1; for (int j = 0; j < NJ; ++j) if (row_moved(r, j)) d = j; return d; }
    static constexpr int deepest2(int r, int s) { int d = -1; for (int j = 0; j < NJ; ++j) if (row_moved(r, j) && row_moved(s, j)) d = j; return d; }
    // is there an EE hinge strictly below j that moves row r's body?  (then DJ[r][j] has been written when hinge j is reached)
    static constexpr bool dj_written(int r, int j) { for (int c = j + 1; c < NJ; ++c) if (TI::above(j, c) && row_moved(r, c)) return true; return false; }
    static constexpr int m_entry(int i, int j) { return i == j ? TI::diag_index(j) : TI::pair_index(i, j); }      // i at or above j
    static constexpr int tri(int r, int s) { return r * (r + 1) / 2 + s; }                                         // s <= r
};

using r16::static_for;
using r16::static_for_down;

// Anchor a value where it is computed.  The recursion below is pure arithmetic between loads; the instruction selector orders pure
// nodes by their USES, and every use sits behind the recursion (the k x k stage) -- left alone, all 330 loads and the M dq sums come
// first and the factorisation afterwards, with everything live in between (1 kB of scratch).  An empty volatile asm is ordered like a
// side effect: what it touches is computed before the next pin of the schedule.
__device__ __forceinline__ void pin(double& v) { asm volatile("" : "+v"(v)); }
// uniform base (scalar registers, constant offsets folded into it / the instruction) + one 32-bit BYTE offset per lane: the address form
// that needs no 64-bit address arithmetic per access -- and leaves the compiler no per-access address to keep (it spilled 91 of them)
// (the base laundered through scalar registers: left visible, the compiler folds base + lane offset into ONE vector address first and
//  then keeps -- and spills -- a 64-bit vector address per constant offset beyond the instruction's 4 KB)
typedef const __attribute__((address_space(1))) double* gcptr;      // (explicitly global: an address that went through an asm statement is
typedef __attribute__((address_space(1))) double* gptr;             //  otherwise a FLAT one, and its loads flat loads)
__device__ __forceinline__ gcptr sgpr_ptr(const double* p) { gcptr q = (gcptr)p; asm volatile("" : "+s"(q)); return q; }
__device__ __forceinline__ gptr sgpr_ptr(double* p) { gptr q = (gptr)p; asm volatile("" : "+s"(q)); return q; }
__device__ __forceinline__ double ld_su(gcptr ubase, const uint32_t voff) {
    return *reinterpret_cast<gcptr>(reinterpret_cast<const __attribute__((address_space(1))) char*>(ubase) + voff);
}
__device__ __forceinline__ double ld_su(const double* ubase, const uint32_t voff) { return ld_su((gcptr)ubase, voff); }
__device__ __forceinline__ void st_su(gptr ubase, const uint32_t voff, const double v) {
    *reinterpret_cast<gptr>(reinterpret_cast<__attribute__((address_space(1))) char*>(ubase) + voff) = v;
}
__device__ __forceinline__ void st_su(double* ubase, const uint32_t voff, const double v) { st_su((gptr)ubase, voff, v); }
using r16::rsq_refined;
using r16::rcp_refined;

// Records of the flagged robots, transposed: [group of 64 records][entry][64] doubles -- what a lane writes and what a lane of the eigen
// pass reads sits next to its neighbours' (slots of a wave's flagged lanes are consecutive: coalesced both ways).  Entries: the lower
// triangle of A (tri(r, c)), w, the entries of J that can be non-zero (row by row, hinges ascending), the robot's index and the mask of
// rows that are padding or exact zero rows (integers in the doubles' bits).
template <class L>
struct Rec {
    static constexpr int K = L::K, NJ = L::NJ;
    static constexpr int NA = K * (K + 1) / 2;
    static constexpr int OW = NA;
    static constexpr int OJ = NA + K;
    static constexpr int jslot(int r, int j) {      // position of J[r][j] among the movable entries (-1: structurally zero)
        if (!L::row_moved(r, j)) return -1;
        int n = 0;
        for (int r2 = 0; r2 < K; ++r2)
            for (int j2 = 0; j2 < NJ; ++j2) {
                if (r2 == r && j2 == j) return n;
                n += L::row_moved(r2, j2) ? 1 : 0;
            }
        return -1;
    }
    static constexpr int n_j() { int n = 0; for (int r = 0; r < K; ++r) for (int j = 0; j < NJ; ++j) n += L::row_moved(r, j) ? 1 : 0; return n; }
    static constexpr int OM = OJ + n_j();
    static constexpr int E = OM + 2;
    static_assert(E <= REC_DOUBLES, "the host allocates REC_DOUBLES per robot");
};

template <class TOPO, class SH, typename TIN>
__global__ __launch_bounds__(64, 1) void osc_lane_kernel(const Row16Train<TIN> tr, const LaneTrain lt) {
    using L = LT<TOPO, SH>;
    using TI = FeTopo<TOPO>;
    constexpr int NJ = L::NJ, K = L::K;
    static_assert(SH::NC == L::n_cand(), "one row count per end-effector candidate of the tree");
    static_assert(K >= 1 && K <= IRLOSC_MAX_K && L::n_ee() <= 16, "shape");
    static_assert(TI::depth_first(), "hinges numbered depth first: a subtree is a run of indices");
    constexpr int BLK_E = TI::n_compact();
    constexpr int ZERO_E = TI::zero_index();
    constexpr int TASK_E = TI::task_index(0);
    constexpr int NE = TI::n_pairs() + NJ;
    const KParams<TIN>& p = tr.p[blockIdx.y];
    const Row16Extra& x = tr.x[blockIdx.y];
    const int lane = threadIdx.x;
    const int b = blockIdx.x * 64 + lane;
    const bool live = b < p.B;
    const int bc = live ? b : p.B - 1;
    const double* __restrict__ col = x.side + (size_t)blockIdx.x * BLK_E * 64 + lane;
    const double* __restrict__ gq = lt.qt[blockIdx.y] + (size_t)blockIdx.x * (2 * NJ * 64) + lane;
    const uint32_t realm = lt.map.real;
    __shared__ double s_u[NJ * 65];                    // M dq as the hinges complete, then the torques: [hinge][64 robots + 1 pad]
    __shared__ double s_w[K * 64];                     // the task vector waits here across the recursion and the k x k stage
    // -DIRLOSC_LANE_STAMPS (tools/build_variant.py; tools/phase_timing.py ... fromq with IRLOSC_PHASE_LANE=1): cycle stamps per phase into
    // p.dbg.  Compile-time only: a run-time `if (p.dbg)` around a stamp is a branch, and a branch inside the recursion lets the compiler
    // sink the arithmetic out of its pinned regions (see pin()).
#ifdef IRLOSC_LANE_STAMPS
    unsigned long long ts[8];
#define LANE_TS(i) ts[i] = __builtin_readcyclecounter()
#else
#define LANE_TS(i)
#endif
    LANE_TS(0);
    const unsigned long long rt0 = x.span ? __builtin_amdgcn_s_memrealtime() : 0ull;      // irlosc_time_trains (see Row16Extra::span)
    const unsigned long long cyc0 = x.span ? (unsigned long long)__builtin_readcyclecounter() : 0ull;

    // The row map as scalars, read ONCE (left where they are used they become scalar loads + branches inside the recursion, which
    // cut it into basic blocks the scheduling pins cannot hold together): rm[r] = 1 for a task row, 0 for padding; a padding row reads
    // the entry of zeros -- entry = ZERO + rm (e0 - ZERO) + rm comp, no branch.
    int rm[K], rmc[K];
#pragma unroll
    for (int r = 0; r < K; ++r) {
        rm[r] = (int)((realm >> r) & 1u);
        rmc[r] = rm[r] * lt.map.comp[r];
    }
    auto jload = [&](auto rc, auto jc) -> double {
        constexpr int r = decltype(rc)::value, j = decltype(jc)::value;
        constexpr int de = L::jentry0(r, j) - ZERO_E;
        const int e = ZERO_E + rm[r] * de + rmc[r];
        return col[(size_t)(unsigned)e * 64];
    };

    // ---- what is uniform over the recursion, up front: gains, null-space gain, the task rows and the wrench (their latency hides
    // behind the first rows of M) -----------------------------------------------------------------------------------------------------
    // (No branch anywhere between here and the end of the k x k stage: with one in between, the compiler SINKS the whole factorisation --
    //  every value that is only used behind the branch -- out of the pinned regions below, and all 330 loads stay live at once.  Hence
    //  loads through selected addresses instead of `if (flag) load`.)
    const bool has_wr = (p.cfgflags & IRLOSC_ADMITTANCE) && p.wrench != nullptr;
    const bool use_g = (p.cfgflags & IRLOSC_USE_G) != 0;
    const int nd = p.ndev;
    const TIN* __restrict__ zeros = reinterpret_cast<const TIN*>(x.zeros);
    const double kvn_in = (double)p.null_kv[p.gains_per_instance ? bc : 0];      // (always allocated; unused without the flag)
    const double kvn = (p.cfgflags & IRLOSC_NULLSPACE) ? kvn_in : 0.0;
    double kvd[IRLOSC_MAX_DEV];
    uint32_t jm[IRLOSC_MAX_DEV];
#pragma unroll
    for (int d2 = 0; d2 < IRLOSC_MAX_DEV; ++d2) {
        const int dd = d2 < nd ? d2 : 0;
        kvd[d2] = (double)p.gains[(p.gains_per_instance ? (size_t)bc * nd * IRLOSC_GAIN_WORDS : 0) + dd * IRLOSC_GAIN_WORDS + 1];
        jm[d2] = d2 < nd ? p.dev[dd].joint_mask : 0u;
    }
    const TIN* __restrict__ wbase = has_wr ? p.wrench : zeros;
    const size_t wsel = has_wr ? 1 : 0;
    double dq[NJ];                  // joint velocities: requested with the first row that needs them
    double Mv[NE];                  // entries of M as they arrive (prefetched PFD hinges ahead)
    double Jv[K][NJ];               // entries of J likewise
    double Bv[NJ];                  // bias forces
    double Dl[NE];                  // accumulated updates, keyed like the entries of M
    double DJ[K][NJ];
    double Al[K * (K + 1) / 2];     // lower triangle of A = Y^T Y
    double macc[NJ];                // M dq: partial sums of the hinges whose row has not been read yet
    double dx[K];
    double lrow[NJ];
    double yv[K];
#pragma unroll
    for (int r = 0; r < K; ++r) dx[r] = 0.0;
#pragma unroll
    for (int e = 0; e < K * (K + 1) / 2; ++e) Al[e] = 0.0;      // (rows of different candidates without a common hinge stay zero)
    bool npd = false;

    auto fetch = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        static_for<0, j + 1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (TI::above(i, j)) {
                constexpr int e = L::m_entry(i, j);
                Mv[e] = col[(size_t)e * 64];
                if constexpr (L::subtree_last(i) == j) dq[i] = gq[(2 * i + 1) * 64];      // row j is the first that multiplies with dq_i
            }
        });
        {   // (the entry of zeros when the bias forces are off: a selected address, no branch)
            constexpr int eb = TI::bias_index(j);
            Bv[j] = col[(size_t)(use_g ? eb : ZERO_E) * 64];
        }
        if constexpr (L::hinge_ee(j)) {
            static_for<0, K>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                if constexpr (L::row_moved(r, j)) Jv[r][j] = jload(rc, jc);
            });
        }
    };
    // Hinges requested ahead of the one being eliminated.  One wave per SIMD: nothing else covers the ~2 us of an HBM round trip, and
    // a hinge is only ~150 instructions -- at a distance of 2 every hinge waited for its row (703 us per train: 21 cycles per
    // instruction).  The prefetched values may sit in the accumulation half of the register file (loads write AGPRs directly).
#ifndef IRLOSC_LANE_PFD
#define IRLOSC_LANE_PFD 6
#endif
    constexpr int PFD = IRLOSC_LANE_PFD;
    static_for<0, PFD>([&](auto dc) { constexpr int j = NJ - 1 - decltype(dc)::value; if constexpr (j >= 0) fetch(std::integral_constant<int, (j >= 0 ? j : 0)>{}); });
#ifndef IRLOSC_LANE_TASK_IN
#define IRLOSC_LANE_TASK_IN 1
#endif
    if constexpr (IRLOSC_LANE_TASK_IN) {
        // Part 1 of the task signal -- calc_error, velocity limit, gains, stiffness (osc.py:101-118,70-99,160-168) [+ the wrench,
        // osc.py:184-185] -- computed HERE, device by device, while the first rows of M are on their way: this wave has nothing else to do
        // until they arrive (one wave per SIMD), so the ~450 instructions per device are free, and the task pass (58 us per train, 230 MB
        // of traffic) is not launched at all.  Same formulas in the same order as osc_task_rows_fromq_kernel (task_rot, atan2,
        // apply_gains6_fast).  The rows go to their CANONICAL positions in LDS (padding rows: zero).
#pragma unroll
        for (int r = 0; r < K; ++r) s_w[r * 64 + lane] = 0.0;
#pragma unroll 1
        for (int dv = 0; dv < nd; ++dv) {
            const DevMeta dm = p.dev[dv];
            const int e0 = lt.map.ee_e0[dv];
            const TIN* __restrict__ tgp = p.tgt + ((size_t)bc * nd + dv) * 7;
            const TIN* __restrict__ gp = p.gains + (p.gains_per_instance ? (size_t)bc * nd * IRLOSC_GAIN_WORDS : 0) + dv * IRLOSC_GAIN_WORDS;
            const TIN* __restrict__ wp = wbase + wsel * (((size_t)bc * nd + dv) * 6);
            double ee[7], tg[7], g[IRLOSC_GAIN_WORDS], wr6[6];
#pragma unroll
            for (int i = 0; i < 7; ++i) ee[i] = col[(size_t)(unsigned)(e0 + i) * 64];
#pragma unroll
            for (int i = 0; i < 7; ++i) tg[i] = (double)tgp[i];
#pragma unroll
            for (int i = 0; i < IRLOSC_GAIN_WORDS; ++i) g[i] = (double)gp[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) wr6[i] = (double)wp[wsel * i];
            double e[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (dm.calc & 1u) { e[0] = ee[0] - tg[0]; e[1] = ee[1] - tg[1]; e[2] = ee[2] - tg[2]; }
            if (dm.calc & 2u) {
                const r16::TaskRot R = r16::task_rot(ee, tg);
#pragma unroll
                for (int a2 = 0; a2 < 3; ++a2) {
                    double ay, ax;
                    R.angle_args(a2, ay, ax);
                    e[3 + a2] = atan2(ay, ax);
                }
            }
            r16::apply_gains6_fast(g, e);
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if (dm.dofmask & (1u << i)) { s_w[lt.map.canon[dm.row0 + cnt] * 64 + lane] = e[i] + wr6[i]; ++cnt; }
        }
    } else
    {   // Part 1 of the task signal [+ the wrench]: u_task_all + ext_f (osc.py:184-185), canonical order (padding: the entry of zeros;
        // device 0, component 0 -- a valid address, times zero).  Requested behind the first rows of M -- one round trip for all of it --
        // and parked in LDS: the values are next needed behind the k x k stage, and any register they held on the way was spilled, load
        // by load, each waited for on the spot (13 to 26 round trips per wave).
        double wt[K], wrv[K];
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int e = ZERO_E + rm[r] * (TASK_E - ZERO_E + lt.map.ext[r]);
            wt[r] = col[(size_t)(unsigned)e * 64];
            wrv[r] = (double)wbase[wsel * (((size_t)bc * nd + lt.map.dev[r]) * 6 + lt.map.comp[r])];
        }
#pragma unroll
        for (int r = 0; r < K; ++r) s_w[r * 64 + lane] = fma((double)rm[r], wrv[r], wt[r]);
    }

    LANE_TS(1);
    // ---- the recursion, hinges NJ - 1 .. 0 --------------------------------------------------------------------------------------
    static_for_down<0, NJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j - PFD >= 0) fetch(std::integral_constant<int, (j - PFD >= 0 ? j - PFD : 0)>{});
        __builtin_amdgcn_sched_barrier(0);
        constexpr int ed = L::m_entry(j, j);
        constexpr bool below = L::has_below(j);
        double d = Mv[ed];
        double mj = d * dq[j];
        if constexpr (below) { mj += macc[j]; d -= Dl[ed]; }
        static_for<0, j>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (TI::above(i, j)) {
                constexpr int e = L::m_entry(i, j);
                const double m = Mv[e];
                mj = fma(m, dq[i], mj);
                if constexpr (L::subtree_last(i) == j) macc[i] = m * dq[j];
                else macc[i] = fma(m, dq[j], macc[i]);
                pin(macc[i]);
                lrow[i] = below ? m - Dl[e] : m;
            }
        });
        {   // (M dq)_j is complete -- every row under j and row j itself have been read: the torque without its task term waits in LDS
            // u0 (branch A damping, assignment in device order: osc.py:174) + bias - kvn (M dq)_j
            double cf = 0.0;
#pragma unroll
            for (int d2 = 0; d2 < IRLOSC_MAX_DEV; ++d2) cf = ((jm[d2] >> j) & 1u) ? -kvd[d2] : cf;
            s_u[j * 65 + lane] = fma(cf - kvn, mj, Bv[j]);
        }
        npd = npd | !(d > 0.0);                                // also catches NaN
        d = fmax(d, 1e-300);
        const double rs = rsq_refined(d);
        static_for<0, j>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (TI::above(i, j)) lrow[i] *= rs;      // L[j][i]
        });
        static_for<0, j>([&](auto ac) {
            constexpr int a = decltype(ac)::value;
            if constexpr (TI::above(a, j)) {
                static_for<0, a + 1>([&](auto bc2) {
                    constexpr int b2 = decltype(bc2)::value;
                    if constexpr (TI::above(b2, j)) {
                        constexpr int e = L::m_entry(b2, a);
                        // the first hinge (in this order) under both: the last index of the deeper one's subtree
                        if constexpr (L::subtree_last(a) == j) Dl[e] = lrow[a] * lrow[b2];
                        else Dl[e] = fma(lrow[a], lrow[b2], Dl[e]);
                        pin(Dl[e]);
                    }
                });
            }
        });
        if constexpr (L::hinge_ee(j)) {
            static_for<0, K>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                if constexpr (L::row_moved(r, j)) {
                    const double jv = Jv[r][j];
                    dx[r] = fma(jv, dq[j], dx[r]);
                    pin(dx[r]);
                    double t = jv;
                    if constexpr (L::dj_written(r, j)) t -= DJ[r][j];
                    const double y = t * rs;                   // Y[j][r]
                    yv[r] = y;
                    static_for<0, j>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        if constexpr (TI::above(i, j)) {
                            if constexpr (L::deepest(r) == j) DJ[r][i] = lrow[i] * y;
                            else DJ[r][i] = fma(lrow[i], y, DJ[r][i]);
                            pin(DJ[r][i]);
                        }
                    });
                }
            });
            static_for<0, K>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                if constexpr (L::row_moved(r, j)) {
                    static_for<0, r + 1>([&](auto sc) {
                        constexpr int s = decltype(sc)::value;
                        if constexpr (L::row_moved(s, j)) {
                            constexpr int e = L::tri(r, s);
                            if constexpr (L::deepest2(r, s) == j) Al[e] = yv[r] * yv[s];
                            else Al[e] = fma(yv[r], yv[s], Al[e]);
                            pin(Al[e]);
                        }
                    });
                }
            });
        }
    });
    __builtin_amdgcn_sched_barrier(0);
#ifndef IRLOSC_LANE_JT_EARLY
#define IRLOSC_LANE_JT_EARLY 0
#endif
    // The Jacobian entries once more, for J^T t behind the k x k stage (and for the records of the flagged robots).  Requested ahead of
    // that stage their round trip would run under its ~2 000 instructions -- but 85 more live doubles next to the factor's 91 spill
    // (1.3 kB of scratch): they are requested behind it.
    double Jt[K][NJ];
    auto load_jt = [&]() {
        static_for<0, NJ>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (L::hinge_ee(j)) {
                static_for<0, K>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    if constexpr (L::row_moved(r, j)) Jt[r][j] = jload(rc, jc);
                });
            }
        });
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (IRLOSC_LANE_JT_EARLY == 1) load_jt();

    LANE_TS(2);
    uint32_t flags = npd ? IRLOSC_FLAG_M_NOT_PD : 0u;
    // ---- w = u_task_all [+ ext_f] - kvn dx (null-space term folded in: osc_generic.hpp header) -------------------------------------
    bool nr[K];                     // row r is padding, or a task row no joint can move (A[r][r] == 0 exactly: row r of J is zero)
    bool anyzero = false;
    static_for<0, K>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const bool real = rm[r] != 0;
        const bool zr = real && Al[L::tri(r, r)] == 0.0;
        anyzero = anyzero | zr;
        nr[r] = !real | zr;
        s_w[r * 64 + lane] = nr[r] ? 0.0 : s_w[r * 64 + lane] - kvn * dx[r];      // (own slot: no other lane reads it)
    });
    // ---- k x k: L~ D L~^T in place (Lf), the certificate, the plain solve (osc.py:51-55; osc_row16.hpp) --------------------------
    double nA2 = 0.0;
#pragma unroll
    for (int r = 0; r < K; ++r) {
#pragma unroll
        for (int s = 0; s <= r; ++s) { const double a = Al[r * (r + 1) / 2 + s]; nA2 = fma(a, r == s ? a : 2.0 * a, nA2); }
    }
    // In place: A itself is not kept (a copy of its 91 entries next to the factor's is 360 registers, and the arithmetic only reaches
    // the 256 architectural ones).  The robots that need A again -- the eigen pass's -- get it back as L~ D' L~^T - diag(d' - d) below:
    // a pivot replaced by 1 is a change of that diagonal entry of A and of nothing else.
    double (&Lf)[K * (K + 1) / 2] = Al;
    double invd[K], dtrue[K];
    bool pd = true;
    double det = 1.0;
    static_for<0, K>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        double d = Lf[L::tri(j, j)];
        dtrue[j] = d;
        const bool bad = !nr[j] & !(d > 0.0);
        pd = pd & !bad;
        d = (bad | nr[j]) ? 1.0 : d;
        det *= d;
        const double iv = rcp_refined(d);
        invd[j] = iv;
        double f[K];
        static_for<j + 1, K>([&](auto ic) { constexpr int i = decltype(ic)::value; f[i] = Lf[L::tri(i, j)] * iv; });
        static_for<j + 1, K>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            static_for<j + 1, i + 1>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                Lf[L::tri(i, c)] = fma(-f[i], Lf[L::tri(c, j)], Lf[L::tri(i, c)]);
            });
        });
        static_for<j + 1, K>([&](auto ic) { constexpr int i = decltype(ic)::value; Lf[L::tri(i, j)] = f[i]; });
        __builtin_amdgcn_sched_barrier(0);      // (column by column: left to itself the scheduler interleaves the whole stage and everything long-lived spills)
    });
    // trace(A^-1) = sum over the columns m of W = L~^-1 of sum_c W[c][m]^2 / d_c (real rows only)
    double trA = 0.0;
    static_for<0, K>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        double xw[K];
        xw[m] = 1.0;
        double acc = nr[m] ? 0.0 : invd[m];
        static_for<m + 1, K>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            double s = -Lf[L::tri(c, m)];
            static_for<m + 1, c>([&](auto qc) { constexpr int q2 = decltype(qc)::value; s = fma(-Lf[L::tri(c, q2)], xw[q2], s); });
            xw[c] = s;
            acc = fma(s * s, nr[c] ? 0.0 : invd[c], acc);
        });
        trA += acc;
        if constexpr (m % 2 == 1) __builtin_amdgcn_sched_barrier(0);      // two columns of W at a time
    });
    const bool small_det = !pd | !(fabs(det) >= 1e-4) | anyzero;
    const double cond_bound = sqrt(nA2) * trA;
    const bool plain = pd & t_finite(cond_bound) & (!small_det | (cond_bound < 0.99e5));
    flags |= small_det ? IRLOSC_FLAG_PINV_BRANCH : 0u;
    flags |= plain ? 0u : IRLOSC_FLAG_EIGEN_PATH;
    flags |= anyzero ? IRLOSC_FLAG_TRUNCATED : 0u;
    double t[K], w[K];
#pragma unroll
    for (int r = 0; r < K; ++r) w[r] = s_w[r * 64 + lane];
    static_for<0, K>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        double s = w[c];
        static_for<0, c>([&](auto qc) { constexpr int q2 = decltype(qc)::value; s = fma(-Lf[L::tri(c, q2)], t[q2], s); });
        t[c] = s;
    });
    static_for<0, K>([&](auto cc) { constexpr int c = decltype(cc)::value; t[c] *= invd[c]; });
    static_for_down<0, K>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        double s = t[c];
        static_for<c + 1, K>([&](auto ic) { constexpr int i = decltype(ic)::value; s = fma(-Lf[L::tri(i, c)], t[i], s); });
        t[c] = s;
    });
    static_for<0, K>([&](auto cc) { constexpr int c = decltype(cc)::value; t[c] = plain ? t[c] : 0.0; });      // the eigen pass adds its own J^T t
    __builtin_amdgcn_sched_barrier(0);
    LANE_TS(3);
    if constexpr (IRLOSC_LANE_JT_EARLY == 2) load_jt();      // (in flight while the records are written)
    // ---- the robots the certificate does not clear get a record for the eigen pass: slot from one atomic per wave ---------------------
    const bool hand = !plain && live;
    const unsigned long long hm = __ballot(hand);
    double* __restrict__ rec = nullptr;
    uint32_t ro = 0u;                  // (transposed records: uniform base + a 32-bit BYTE offset per lane -- one address register, not one per store)
    if (hm != 0ull) {
        const int first = (int)__builtin_ctzll(hm);
        int base = 0;
        if (lane == first) base = atomicAdd(lt.rec_count[blockIdx.y], (int)__builtin_popcountll(hm));
        base = __builtin_amdgcn_readlane(base, first);
        const int slot = base + (int)__builtin_popcountll(hm & ((1ull << lane) - 1ull));
        rec = lt.rec[blockIdx.y];
        ro = ((uint32_t)((hand ? slot : 0) >> 6) * (uint32_t)(Rec<L>::E * 64) + (uint32_t)((hand ? slot : 0) & 63)) * 8u;
    }
    if (hand) {
        uint32_t nrm = 0u;
        static_for<0, K>([&](auto rc) { constexpr int r = decltype(rc)::value; nrm |= nr[r] ? (1u << r) : 0u; });
        static_for<0, K>([&](auto rc) { constexpr int eo = (Rec<L>::OW + decltype(rc)::value) * 64; st_su(rec + eo, ro, w[decltype(rc)::value]); });
        st_su(rec + Rec<L>::OM * 64, ro, __builtin_bit_cast(double, (long long)b));
        st_su(rec + (Rec<L>::OM + 1) * 64, ro, __builtin_bit_cast(double, (long long)nrm));
        // A = L~ D' L~^T - diag(d' - d), row by row and straight into the record (no second triangle in registers: with the factor's 91
        // entries that is what spilled): Ldr[j] = L~[r][j] d'_j, A[r][c] = sum_{j < c} Ldr[j] L~[c][j] + Ldr[c], the true pivot on the diagonal
        double dpr[K];
        static_for<0, K>([&](auto jc) { constexpr int j = decltype(jc)::value; dpr[j] = (dtrue[j] > 0.0 && !nr[j]) ? dtrue[j] : 1.0; });
        static_for<0, K>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            double Ldr[K];
            static_for<0, r>([&](auto jc) { constexpr int j = decltype(jc)::value; Ldr[j] = Lf[L::tri(r, j)] * dpr[j]; });
            double arow[K];                  // the lower triangle's row r
            static_for<0, r + 1>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                double acc;
                if constexpr (c < r) acc = Ldr[c]; else acc = dtrue[r];
                static_for<0, c>([&](auto jc) { constexpr int j = decltype(jc)::value; acc = fma(Ldr[j], Lf[L::tri(c, j)], acc); });
                arow[c] = acc;
            });
            static_for<0, r + 1>([&](auto cc) { constexpr int c = decltype(cc)::value; constexpr int eo = L::tri(r, c) * 64; st_su(rec + eo, ro, arow[c]); });
        });
    }
    __builtin_amdgcn_sched_barrier(0);
    LANE_TS(4);
    // ---- J^T t, hinge by hinge, behind the records' A (the factor is dead by now: its 91 entries and these 85 do not fit the
    // architectural registers together); the flagged robots leave the entries that can be non-zero in their record
    if constexpr (IRLOSC_LANE_JT_EARLY == 0) load_jt();
    double jt[NJ];
    static_for<0, NJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (L::hinge_ee(j)) {
            double s = 0.0;
            static_for<0, K>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                if constexpr (L::row_moved(r, j)) s = fma(Jt[r][j], t[r], s);
            });
            jt[j] = s;
        }
    });
    if (hand) {
        static_for<0, K>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            static_for<0, NJ>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (L::row_moved(r, j)) { constexpr int eo = (Rec<L>::OJ + Rec<L>::jslot(r, j)) * 64; st_su(rec + eo, ro, Jt[r][j]); }
            });
        });
    }
    LANE_TS(5);
    // ---- torques (osc.py:174,184-200): u = (u0 + bias - kvn M dq, parked in LDS hinge by hinge) - J^T t -----------------------------------
    bool bad = false;
    static_for<0, NJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        double u0 = s_u[j * 65 + lane];
        if constexpr (L::hinge_ee(j)) { u0 -= jt[j]; s_u[j * 65 + lane] = u0; }
        bad = bad | !t_finite(u0);
    });
    flags |= bad ? IRLOSC_FLAG_NONFINITE : 0u;
    if (live) p.flags[b] = flags;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    {
        const int nrob = min(64, p.B - (int)blockIdx.x * 64);
        TIN* __restrict__ uo = p.u + (size_t)blockIdx.x * 64 * NJ;
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const int idx = lane + 64 * i;
            const int rob = idx / NJ, jn = idx - rob * NJ;
            if (rob < nrob) uo[idx] = (TIN)s_u[jn * 65 + rob];
        }
    }
#ifdef IRLOSC_LANE_STAMPS
    LANE_TS(6);
    if (p.dbg && lane == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) p.dbg[(size_t)blockIdx.x * 10 + i] = ts[i];
        p.dbg[(size_t)blockIdx.x * 10 + 7] = ts[6];
        p.dbg[(size_t)blockIdx.x * 10 + 8] = 0;
        p.dbg[(size_t)blockIdx.x * 10 + 9] = 0;
    }
#endif
#undef LANE_TS
    if (x.span && lane == 0) {       // first wave's start / last wave's end of the train, untraced (irlosc_time_trains)
        unsigned long long* sp = x.span + 2 * (blockIdx.x & (R16_SPAN_SLOTS - 1));
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        atomicMin(sp, rt0);
        atomicMax(sp + 1, rt1);
        if (blockIdx.x == 0 && blockIdx.y == 0) {
            x.span[2 * R16_SPAN_SLOTS] = (unsigned long long)__builtin_readcyclecounter() - cyc0;
            x.span[2 * R16_SPAN_SLOTS + 1] = rt1 - rt0;
        }
    }
}


// The eigen pass behind the lane kernel, ONE LANE PER FLAGGED ROBOT: 64 records per wave (transposed records: every load and store of the
// record is a run of neighbouring words), the algorithm of r16::eigen16 (osc_row16.hpp: t = pinv(A, rcond 1e-5) w by deflated inverse
// iteration through L~ D L~^T, Rayleigh-Ritz on several candidates, the cut decided by the inertia of (1e5 theta) I - A; osc.py:55)
// restated on per-lane arrays -- a dot product is thirteen FMAs of a lane instead of a 16-lane butterfly, a solve is the same chain of
// FMAs without the broadcasts -- so 64 robots cost ~13 k instructions where four-per-wave cost 16 x 2.5 k.  Every decision is per lane
// and frozen at the lane's own convergence (branches are wave-uniform `__any` tests around masked updates): a robot's result does not
// depend on its wave-mates.  A is not held across the iteration (the factor's 91 doubles, four candidate vectors and the iterate are
// the register file): the rare blocks that need it again -- Rayleigh-Ritz, the inertia tests, the shifted refactorisation -- reload it
// from the record and factor once more behind them.  Then u -= J^T t on the torques the lane kernel left without the task term.
template <class TOPO, class SH, typename TIN>
__global__ __launch_bounds__(64, 1) void osc_lane_eigen_kernel(const EigTrain et) {
    using L = LT<TOPO, SH>;
    using R = Rec<L>;
    constexpr int NJ = L::NJ, K = L::K, NA = R::NA;
    constexpr int NV = 4;
    const EigStep& es = et.s[blockIdx.y];
    const int n = __builtin_amdgcn_readfirstlane(min(*es.rec_count, et.B));
    if (n < et.lane_min) return;            // (a thin list: osc_lane_eigen16_kernel below has this step)
    const int lane = threadIdx.x;
    for (int g = blockIdx.x; g * 64 < n; g += gridDim.x) {
        const bool live = g * 64 + lane < n;
        const double* __restrict__ rg = es.rec + (size_t)g * (R::E * 64);      // (uniform; the lane adds a 32-bit word offset)
        const uint32_t lo32 = (uint32_t)(live ? lane : n - 1 - g * 64);
        const uint32_t vo = lo32 * 8u;
        const long long bid = __builtin_bit_cast(long long, ld_su(rg + R::OM * 64, vo));
        const uint32_t zrow = (uint32_t)__builtin_bit_cast(long long, ld_su(rg + (R::OM + 1) * 64, vo));
        bool nr[K];
#pragma unroll
        for (int r = 0; r < K; ++r) nr[r] = ((zrow >> r) & 1u) != 0;
        double Lf[NA], invd[K];
        auto load_A = [&](double (&dst)[NA]) {
            static_for<0, (NA + 7) / 8>([&](auto bc2) {
                constexpr int e0 = 8 * decltype(bc2)::value;
                const gcptr bp = sgpr_ptr(rg + e0 * 64);              // (eight entries within the instruction's offset field)
                static_for<0, 8>([&](auto ec) { constexpr int e = e0 + decltype(ec)::value; if constexpr (e < NA) dst[e] = ld_su(bp + (e - e0) * 64, vo); });
            });
        };
        // in place: L~ below the diagonal, pivots not kept (invd); a pivot of a padded / zero row, or a non-positive one, is taken as 1
        auto factor = [&](double (&F)[NA], const double sigma, bool& pd) {
            pd = true;
            static_for<0, K>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                double d = F[L::tri(j, j)] + sigma;
                const bool npd = !nr[j] && !(d > 0.0);
                pd = pd && !npd;
                d = (npd || nr[j]) ? 1.0 : d;
                const double iv = rcp_refined(d);
                invd[j] = iv;
                double f[K];
                static_for<j + 1, K>([&](auto ic) { constexpr int i = decltype(ic)::value; f[i] = F[L::tri(i, j)] * iv; });
                static_for<j + 1, K>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    static_for<j + 1, i + 1>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        F[L::tri(i, c)] = fma(-f[i], F[L::tri(c, j)], F[L::tri(i, c)]);
                    });
                });
                static_for<j + 1, K>([&](auto ic) { constexpr int i = decltype(ic)::value; F[L::tri(i, j)] = f[i]; });
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        auto solve = [&](double (&z)[K]) {      // z <- (L~ D L~^T)^-1 z: the FMA chains of r16::solve16, per lane
            static_for<0, K - 1>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                static_for<j + 1, K>([&](auto cc) { constexpr int c = decltype(cc)::value; z[c] = fma(-z[j], Lf[L::tri(c, j)], z[c]); });
            });
#pragma unroll
            for (int c = 0; c < K; ++c) z[c] *= invd[c];
            static_for_down<1, K>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                static_for<0, i>([&](auto cc) { constexpr int c = decltype(cc)::value; z[c] = fma(-z[i], Lf[L::tri(i, c)], z[c]); });
            });
        };
        auto dot = [&](const double (&a)[K], const double (&b2)[K]) -> double {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int c = 0; c < K; ++c) { if (c & 1) s1 = fma(a[c], b2[c], s1); else s0 = fma(a[c], b2[c], s0); }
            return s0 + s1;
        };
        load_A(Lf);
        double nA2 = 0.0, lo = 0.0;
#pragma unroll
        for (int r = 0; r < K; ++r) {
#pragma unroll
            for (int c = 0; c <= r; ++c) { const double a = Lf[L::tri(r, c)]; nA2 = fma(a, r == c ? a : 2.0 * a, nA2); }
            lo = fmax(lo, Lf[L::tri(r, r)]);
        }
        bool pdA;
        factor(Lf, 0.0, pdA);
        double trA = 0.0;      // trace(A^-1) over the real rows: columns of L~^-1
        static_for<0, K>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            double xw[K];
            xw[m] = 1.0;
            double acc = nr[m] ? 0.0 : invd[m];
            static_for<m + 1, K>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                double s2 = -Lf[L::tri(c, m)];
                static_for<m + 1, c>([&](auto qc) { constexpr int q2 = decltype(qc)::value; s2 = fma(-Lf[L::tri(c, q2)], xw[q2], s2); });
                xw[c] = s2;
                acc = fma(s2 * s2, nr[c] ? 0.0 : invd[c], acc);
            });
            trA += acc;
            if constexpr (m % 2 == 1) __builtin_amdgcn_sched_barrier(0);
        });
        // ---- r16::eigen16, per lane ------------------------------------------------------------------------------------------------
        const double hi = sqrt(nA2);
        double sigma = 0.0;
        bool giveup = live && !(hi > 0.0 && t_finite(hi));
        const bool broken = !pdA || !(trA * hi < 1e11);
        if (__any(live && broken)) {          // (that robot with sigma = 2^-40 ||A||_F on the diagonal, the others with 0: their factors again)
            const bool use = live && broken;
            sigma = use ? hi * 0x1p-40 : 0.0;
            bool pd2;
            load_A(Lf);
            factor(Lf, sigma, pd2);
            giveup = giveup || (use && !pd2);
        }
        lo = (lo > 0.0 && lo <= hi) ? lo : hi * 0.25;
        const double net = 4e-5 * hi;
        double v[NV][K], th[NV] = {0.0, 0.0, 0.0, 0.0};
        bool has[NV] = {false, false, false, false};
#pragma unroll
        for (int i = 0; i < NV; ++i) {
#pragma unroll
            for (int c = 0; c < K; ++c) v[i][c] = 0.0;
        }
        int m = 0;
        bool active = live && !giveup;
        double rem = (pdA && sigma == 0.0 && trA > 0.0) ? trA : -1.0;
        bool go = true;
        static_for<0, NV>([&](auto sc) {
            constexpr int slot = decltype(sc)::value;
            go = go && __any(active);
            if (go) {
                double x[K];
#pragma unroll
                for (int c = 0; c < K; ++c) x[c] = nr[c] ? 0.0 : 0.3 + 0.1 * (double)(((c + 3 * slot) * 5) % 7) - 0.05 * (double)slot;
                double lam = 0.0, lam_prev = -1.0;
                bool fin = !active;
                auto deflate = [&](double (&y)[K]) {
                    static_for<0, slot>([&](auto s0c) {
                        constexpr int s0 = decltype(s0c)::value;
                        const double pr = -dot(v[s0], y);
#pragma unroll
                        for (int c = 0; c < K; ++c) y[c] = fma(pr, v[s0][c], y[c]);
                    });
                };
                for (int it = 0; it < IRLOSC_EIG_MAXIT; ++it) {
                    double xn[K];
#pragma unroll
                    for (int c = 0; c < K; ++c) xn[c] = x[c];
                    deflate(xn);
                    solve(xn);
                    const double n2 = dot(xn, xn);
                    const double rn = rsq_refined(n2 > 0.0 ? n2 : 1.0);
                    const double lamn = rn - sigma;
                    const bool settled = fabs(lamn - lam_prev) <= 1e-10 * fabs(lamn) || lamn > 4.0 * net;
#pragma unroll
                    for (int c = 0; c < K; ++c) x[c] = fin ? x[c] : xn[c] * rn;
                    lam = fin ? lam : lamn;
                    lam_prev = lam;
                    fin = fin || (it >= IRLOSC_EIG_FLOOR && settled);
                    if (!__any(!fin)) break;
                }
#pragma unroll
                for (int ex = 0; ex < IRLOSC_EIG_EXTRA; ++ex) {
                    double xn[K];
#pragma unroll
                    for (int c = 0; c < K; ++c) xn[c] = x[c];
                    deflate(xn);
                    solve(xn);
                    const double n2 = dot(xn, xn);
                    const double rn = rsq_refined(n2 > 0.0 ? n2 : 1.0);
                    const bool ok = n2 > 0.0 && t_finite(rn);
#pragma unroll
                    for (int c = 0; c < K; ++c) x[c] = ok ? xn[c] * rn : x[c];
                    lam = (ok && lam <= 4.0 * net) ? rn - sigma : lam;
                }
                const bool cand = active && (lam <= net);
                deflate(x);
                {
                    const double n2 = dot(x, x);
                    const double rn = rsq_refined(n2 > 0.0 ? n2 : 1.0);
#pragma unroll
                    for (int c = 0; c < K; ++c) v[slot][c] = cand ? x[c] * rn : 0.0;
                }
                th[slot] = cand ? lam : 0.0;
                has[slot] = cand;
                m += cand ? 1 : 0;
                rem = (cand && rem > 0.0 && lam > 0.0) ? rem - rcp_refined(lam) : (cand ? -1.0 : rem);
                const bool exhausted = rem > 1e-7 * trA && rem * net < 1.0;
                active = cand && !exhausted;
            }
        });
        giveup = giveup || (m == NV);
        // which candidates sit inside the bracket of the cut (decided by inertia below)
        bool below[NV], ask[NV], anyask = false;
#pragma unroll
        for (int i = 0; i < NV; ++i) { below[i] = false; ask[i] = false; }
        const bool rr = __any(m >= 2);
        // (without Rayleigh-Ritz the Ritz values are final here; with it they change below, so the bracket test follows it)
        if (rr) {
            // Rayleigh-Ritz on the span of the candidates: H = V^T A V (4 x 4; unused vectors are zero), cyclic Jacobi, rotations applied to V
            double (&At)[NA] = Lf;              // (the factor is rebuilt behind this block: its registers hold A meanwhile)
            double h[NV][NV];
            load_A(At);
            static_for<0, NV>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                double av[K];
                static_for<0, K>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    double s2 = 0.0;
                    static_for<0, K>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        s2 = fma(At[r >= c ? L::tri(r, c) : L::tri(c, r)], v[j][c], s2);
                    });
                    av[r] = s2;
                });
                static_for<0, j + 1>([&](auto ic) { constexpr int i = decltype(ic)::value; h[i][j] = dot(v[i], av); h[j][i] = h[i][j]; });
            });
            for (int sweep = 0; sweep < 6; ++sweep) {
                bool turned = false;
#pragma unroll
                for (int p2 = 0; p2 < NV - 1; ++p2) {
#pragma unroll
                    for (int q2 = p2 + 1; q2 < NV; ++q2) {
                        const double hpq = h[p2][q2], hpp = h[p2][p2], hqq = h[q2][q2];
                        const bool rot = has[p2] && has[q2] && fabs(hpq) > 1e-300 && fabs(hpq) > 1e-18 * (fabs(hpp) + fabs(hqq));
                        turned = turned || rot;
                        double theta = (hqq - hpp) * rcp_refined(rot ? 2.0 * hpq : 1.0);
                        theta = fmin(fmax(theta, -1e100), 1e100);
                        const double tq = (theta >= 0.0 ? 1.0 : -1.0) * rcp_refined(fabs(theta) + r16::sqrt_fast(theta * theta + 1.0));
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
#pragma unroll
                        for (int c = 0; c < K; ++c) {
                            const double vp = v[p2][c], vq = v[q2][c];
                            v[p2][c] = cs * vp - sn * vq;
                            v[q2][c] = sn * vp + cs * vq;
                        }
                    }
                }
                if (!__any(turned)) break;
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) th[i] = (m >= 2 && has[i]) ? h[i][i] - 0.0 : th[i];
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            below[i] = has[i] && th[i] <= 1e-5 * lo;
            ask[i] = has[i] && !below[i] && th[i] <= 1e-5 * hi;
            anyask = anyask || ask[i];
        }
        // The pinv cut (osc.py:55): at or under 1e-5 lo is cut, over 1e-5 hi is kept, in between "theta <= 1e-5 lambda_max" is "(1e5
        // theta) I - A is not positive definite" -- one L D L^T of that matrix in the Lf registers (the factor is rebuilt behind it).
        bool cut[NV];
        const bool wave_ask = __any(anyask);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            cut[i] = below[i];
            if (__any(ask[i])) {
                load_A(Lf);
#pragma unroll
                for (int e = 0; e < NA; ++e) Lf[e] = -Lf[e];
                bool pdt;
                factor(Lf, ask[i] ? th[i] * 1e5 : 4.0 * hi, pdt);
                cut[i] = cut[i] || (ask[i] && !pdt);
            }
        }
        if (rr || wave_ask) {
            bool pd3;
            load_A(Lf);
            factor(Lf, sigma, pd3);
        }
        int ncut = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
#pragma unroll
            for (int c = 0; c < K; ++c) v[i][c] = cut[i] ? v[i][c] : 0.0;
            ncut += cut[i] ? 1 : 0;
        }
        // t = P (A + sigma)^-1 P w
        // (the tail's loads go through an opaque copy of the lane offset: nothing of the record is requested ahead of the iteration and
        //  carried through it in registers)
        uint32_t vo2 = vo;
        asm volatile("" : "+v"(vo2));
        double tt[K];
#pragma unroll
        for (int c = 0; c < K; ++c) tt[c] = ld_su(sgpr_ptr(rg + (R::OW + c) * 64), vo2);
        bool lv[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) lv[i] = __any(cut[i]);
        auto project = [&]() {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (lv[i]) {
                    const double pr = -dot(v[i], tt);
#pragma unroll
                    for (int c = 0; c < K; ++c) tt[c] = fma(pr, v[i][c], tt[c]);
                }
            }
        };
        project();
        solve(tt);
        project();
        const uint32_t f2 = ncut > 0 ? IRLOSC_FLAG_TRUNCATED : 0u;
        // ---- u -= J^T t ---------------------------------------------------------------------------------------------------------------
        bool bad = false;
        TIN* __restrict__ ub = reinterpret_cast<TIN*>(es.u) + (size_t)bid * NJ;
        static_for<0, NJ>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (L::hinge_ee(j)) {
                double s2 = 0.0;
                static_for<0, K>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    if constexpr (L::row_moved(r, j)) { constexpr int eo = (R::OJ + R::jslot(r, j)) * 64; s2 = fma(ld_su(sgpr_ptr(rg + eo), vo2), tt[r], s2); }
                });
                if (live) {
                    const double u = (double)ub[j] - s2;
                    bad = bad || !t_finite(u);
                    ub[j] = (TIN)u;
                }
            }
        });
        if (live) {
            uint32_t fl = es.flags[bid] | f2;
            if (bad) fl |= IRLOSC_FLAG_NONFINITE;
            es.flags[bid] = fl;
            if (giveup) es.worklist[atomicAdd(es.workcount, 1)] = (int32_t)bid;
        }
    }
}


// The same pass for a THIN list of flagged robots (fewer than EigTrain::lane_min in the step: small batches, layouts whose task spaces
// seldom degenerate): FOUR records per wave in the row16 layout (16 lanes per robot, lane c = column c of A) -- the factorisation, the
// certificate's numbers and r16::eigen16 exactly as the row16 kernel runs them in place (osc_row16.hpp).  A wave of the lane-form pass
// above is one chain of ~16 k instructions (~70 us) whatever it holds; a step with 600 flagged robots is ten such waves on a machine of
// 1 024 SIMDs, while this form spreads them over 150 short waves (measured: 4 096 robots, k13, from_q 3.9e8 with the lane form alone,
// 4.8e8 with this one).  Both kernels are launched behind every lane kernel; each looks at the step's count and one of them returns.
// Reads the transposed records (16 lanes of a robot read 16 different entries: strided, but the list is thin by construction).
template <class TOPO, class SH, typename TIN>
__global__ __launch_bounds__(64, 2) void osc_lane_eigen16_kernel(const EigTrain et) {
    using namespace r16;
    using L = LT<TOPO, SH>;
    using R = Rec<L>;
    constexpr int NJ = L::NJ, K = L::K;
    const EigStep& es = et.s[blockIdx.y];
    const int n = __builtin_amdgcn_readfirstlane(min(*es.rec_count, et.B));
    if (n >= et.lane_min) return;           // (the lane-form pass has this step)
    const int lane = threadIdx.x, q = lane >> 4, l = lane & 15;
    constexpr int NEE = L::n_ee();
    for (int g = blockIdx.x; g * 4 < n; g += gridDim.x) {
        const int ri = g * 4 + q;
        const bool live = ri < n;
        const int sl = live ? ri : n - 1;
        const double* __restrict__ rec = es.rec + (size_t)(sl >> 6) * (R::E * 64) + (sl & 63);      // entry e of this robot: rec[e * 64]
        const bool inl = l < K;                // (lanes K .. 15 of a robot hold the zero columns of the padded 16 x 16 problem)
        const int lc = inl ? l : 0;
        double Ac[K], A[K];
#pragma unroll
        for (int r = 0; r < K; ++r) { const double a = rec[(size_t)(r >= lc ? L::tri(r, 0) + lc : L::tri(lc, 0) + r) * 64]; Ac[r] = inl ? a : 0.0; }
        const double w0 = rec[(size_t)(R::OW + lc) * 64];
        const double w = inl ? w0 : 0.0;
        const uint32_t zrow = (uint32_t)__builtin_bit_cast(long long, rec[(size_t)(R::OM + 1) * 64]);
        double nA2 = 0.0;
#pragma unroll
        for (int r = 0; r < K; ++r) { nA2 = fma(Ac[r], Ac[r], nA2); A[r] = Ac[r]; }
        nA2 = row_sum(nA2);
        double F[K], G[K];
        double invd_own = 0.0, detA = 1.0;
        bool pdA = true;
        ldl16<K, true>(A, l, 0.0, F, G, invd_own, pdA, detA, K, zrow);
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
        trA = (l < K && !((zrow >> l) & 1u)) ? trA : 0.0;
        trA = row_sum(trA * invd_own);
        double t = 0.0;
        uint32_t f2 = 0;
        bool giveup = false;
        eigen16<K, true>(Ac, F, G, invd_own, pdA, nA2, trA, w, l, live, t, f2, giveup, K, zrow);
        // (everything the tail needs is rebuilt from an opaque copy of the lane id: nothing rides through the eigen stage in registers)
        int lane2 = threadIdx.x;
        asm volatile("" : "+v"(lane2));
        const int q2 = lane2 >> 4, l2 = lane2 & 15;
        const int ri2 = g * 4 + q2;
        const bool live2 = ri2 < n;
        const int sl2 = live2 ? ri2 : n - 1;
        const double* __restrict__ rec2 = et.s[blockIdx.y].rec + (size_t)(sl2 >> 6) * (R::E * 64) + (sl2 & 63);
        const long long bid = __builtin_bit_cast(long long, rec2[(size_t)R::OM * 64]);
        // lane l2 of a robot takes the EE hinge of rank l2: its column of J, entry by entry (-1: structurally zero)
        int hinge = 0, je[K];
#pragma unroll
        for (int r = 0; r < K; ++r) je[r] = -1;
        static_for<0, NJ>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (L::hinge_ee(j)) {
                constexpr int cr = L::ee_rank(j);
                hinge = (l2 == cr) ? j : hinge;
                static_for<0, K>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    if constexpr (L::row_moved(r, j)) { constexpr int eo = R::OJ + R::jslot(r, j); je[r] = (l2 == cr) ? eo : je[r]; }
                });
            }
        });
        double jr[K];
#pragma unroll
        for (int r = 0; r < K; ++r) { const double a = rec2[(size_t)(je[r] >= 0 ? je[r] : 0) * 64]; jr[r] = je[r] >= 0 ? a : 0.0; }
        double jt = 0.0;
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, K>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            if constexpr (r == 0) fmac_bc_nop<r>(jt, t, jr[r]);
            else fmac_bc<r>(jt, t, jr[r]);
        });
        __builtin_amdgcn_sched_barrier(0);
        bool bad = false;
        if (live2 && l2 < NEE) {
            TIN* up = reinterpret_cast<TIN*>(et.s[blockIdx.y].u) + (size_t)bid * NJ + hinge;
            const double u = (double)*up - jt;
            bad = !t_finite(u);
            *up = (TIN)u;
        }
        const unsigned long long bm = __ballot(bad);
        if (live2 && l2 == 0) {
            const EigStep& e2 = et.s[blockIdx.y];
            uint32_t fl = e2.flags[bid] | f2;
            if ((bm >> (q2 * 16)) & 0xffffull) fl |= IRLOSC_FLAG_NONFINITE;
            e2.flags[bid] = fl;
            if (giveup) e2.worklist[atomicAdd(e2.workcount, 1)] = (int32_t)bid;
        }
    }
}

}  // namespace lane
}  // namespace irlosc
