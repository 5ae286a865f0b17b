// Generic OSC kernel: ONE WAVEFRONT PER ROBOT INSTANCE, all tiles in LDS, runtime n <= 32, k <= 16.
//
// This is the robust path: it serves any layout, fp32 or fp64, and it is the only kernel that
// carries the k x k symmetric eigen-decomposition needed to honour the reference's
// pseudo-inverse semantics (osc.py:51-55) when the task-space inertia is ill-conditioned.  The
// throughput path (osc_group.hpp) routes such instances here through a worklist.
//
// Algorithm per instance (osc.py:41-200), with the algebra collapsed so that no
// n x n inverse and no n x n null-space projector is ever formed:
//   Mdq = M dq                                        (uv_all, osc.py:151; u_null = -kvn*Mdq, :197)
//   M = L L^T (Cholesky);  Y = L^-1 J^T;  A = Y^T Y   (= J M^-1 J^T = Mx_inv, osc.py:49-50)
//   w = u_task_all [+ ext_f] - kvn * (J dq)           (task signal + null-space term folded:
//        N u_null = u_null - J^T Mx J M^-1 M(-kvn dq) = u_null + kvn J^T Mx (J dq), osc.py:195-200)
//   t = Mx w  with Mx = A^-1 if |det A| >= 1e-4 else pinv(A, rcond 1e-5)      (osc.py:51-55)
//   u = u0 + bias - kvn*Mdq - J^T t                   (osc.py:174,184-191,200)
// A^-1 w is a Cholesky solve whenever that is certifiably equal to the reference's branch:
// either |det| >= 1e-4, or cond(A) <= ||A||_F * ||L_A^-1||_F^2 < 1e5 (no singular value can fall
// under the 1e-5 cut).  Otherwise a cyclic Jacobi eigen-decomposition applies the cut exactly.
#pragma once
#include "osc_common.hpp"

namespace irlosc {

// In-place lower Cholesky of the leading m x m block of A (row stride ld) by one wave.
// Returns false if a pivot was not positive (factorisation then continues on |pivot| so that the
// kernel never produces NaN by itself).  *det receives the product of squared pivots.
template <typename T>
__device__ __forceinline__ bool wave_cholesky(T* A, int m, int ld, int lane, T* det) {
    bool ok = true;
    T dacc = T(1);
    for (int j = 0; j < m; ++j) {
        T d = A[j * ld + j];
        if (!(d > T(0))) {
            ok = false;
            d = (d == d && d != T(0)) ? t_abs(d) : T(1);
        }
        dacc *= d;
        T r = t_sqrt(d);
        T inv = T(1) / r;
        __syncthreads();
        if (lane > j && lane < m) A[lane * ld + j] *= inv;
        if (lane == j) A[j * ld + j] = r;
        __syncthreads();
        int i = j + 1 + (lane & 31);
        int h = lane >> 5;
        if (i < m) {
            T lij = A[i * ld + j];
            for (int c = j + 1 + h; c <= i; c += 2) A[i * ld + c] -= lij * A[c * ld + j];
        }
        __syncthreads();
    }
    *det = dacc;
    return ok;
}

// One instance, one wavefront (a 64-thread block): the whole of osc.py:41-200 for instance b.  `smem` =
// generic_smem_bytes() bytes of LDS.  P is KParams<T> or a reference to one in the constant address space.
template <typename T, typename P>
__device__ __forceinline__ void generic_instance(const P& p, const int b, T* smem) {
    const int lane = threadIdx.x;
    const int n = p.n, k = p.k, ndev = p.ndev;
    const int ldn = n | 1, ldk = k | 1;
    {

    T* Ms = smem;                 // n x ldn   (M, then its Cholesky factor L in the lower triangle)
    T* Js = Ms + n * ldn;         // k x ldn
    T* Ys = Js + k * ldn;         // k x ldn   (row r = L^-1 J_r^T)
    T* As = Ys + k * ldn;         // k x ldk   (A = J M^-1 J^T, kept intact for the eigen path)
    T* Ls = As + k * ldk;         // k x ldk   (Cholesky factor of A)
    T* Ws = Ls + k * ldk;         // k x ldk   (L_A^-1, or the Jacobi eigenvectors V)
    T* dqs = Ws + k * ldk;        // n
    T* mdq = dqs + n;             // n
    T* dxs = mdq + n;             // k
    T* wk = dxs + k;              // k
    T* zk = wk + k;               // k
    T* tk = zk + k;               // k
    int* brA = reinterpret_cast<int*>(tk + k);  // ndev: 1 = damping branch A, 0 = branch B

    const auto* Mg = p.M + (size_t)b * n * n;      // storage type of the records may be narrower than T (mixed path)
    const auto* Jg = p.J + (size_t)b * k * n;
    for (int e = lane; e < n * n; e += 64) Ms[(e / n) * ldn + (e % n)] = Mg[e];
    for (int e = lane; e < k * n; e += 64) Js[(e / n) * ldn + (e % n)] = Jg[e];
    if (lane < n) dqs[lane] = p.dq[(size_t)b * n + lane];
    __syncthreads();

    // ---- Mdq = M dq (osc.py:151) and dx = J dq (osc.py:150) --------------------------------------
    if (lane < n) {
        T s = T(0);
        for (int j = 0; j < n; ++j) s += Ms[lane * ldn + j] * dqs[j];
        mdq[lane] = s;
    }
    if (lane >= 32 && lane - 32 < k) {
        int r = lane - 32;
        T s = T(0);
        for (int j = 0; j < n; ++j) s += Js[r * ldn + j] * dqs[j];
        dxs[r] = s;
    }
    __syncthreads();

    uint32_t flags = 0;
    const auto* gbase = p.gains + (p.gains_per_instance ? (size_t)b * ndev * IRLOSC_GAIN_WORDS : 0);
    const T kvn = (p.cfgflags & IRLOSC_NULLSPACE) ? p.null_kv[p.gains_per_instance ? b : 0] : T(0);

    // ---- per-device task-space signal (osc.py:156-181): lane d handles device d -----------------
    if (lane < ndev) {
        const DevMeta dm = pod_copy<DevMeta>(p.dev[lane]);
        T g[IRLOSC_GAIN_WORDS], ee[7], tg[7];
#pragma unroll
        for (int i = 0; i < IRLOSC_GAIN_WORDS; ++i) g[i] = gbase[lane * IRLOSC_GAIN_WORDS + i];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            ee[i] = p.ee[((size_t)b * ndev + lane) * 7 + i];
            tg[i] = p.tgt[((size_t)b * ndev + lane) * 7 + i];
        }
        T e[6];
        task_error6<T>(ee, tg, dm.calc & 1u, dm.calc & 2u, e);
        apply_gains6<T>(g, e);
        T tv[6];
        bool all_nonzero = p.tvel != nullptr;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            tv[i] = p.tvel ? p.tvel[((size_t)b * ndev + lane) * 6 + i] : T(0);
            all_nonzero = all_nonzero && (tv[i] != T(0));
        }
        // np.all(target_vel) == 0  (osc.py:173): branch A unless ALL six components are non-zero
        brA[lane] = all_nonzero ? 0 : 1;
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (dm.dofmask & (1u << i)) {
                T v = e[i];
                if (all_nonzero) {  // branch B (osc.py:176-177)
                    int row = dm.jidx0 + cnt;
                    T dxv = (row < k) ? dxs[row] : T(0);
                    T damp = (i < 3) ? g[6 + i] : T(1);
                    v += g[1] * (dxv - tv[i]) * damp;
                }
                if ((p.cfgflags & IRLOSC_ADMITTANCE) && p.wrench)
                    v += p.wrench[((size_t)b * ndev + lane) * 6 + i];  // ext_f (osc.py:179-185)
                wk[dm.row0 + cnt] = v;
                ++cnt;
            }
        }
    }
    __syncthreads();
    for (int d = 0; d < ndev; ++d) {
        if (!brA[d]) {
            flags |= IRLOSC_FLAG_VEL_BRANCH_B;
            if (p.dev[d].jidx0 + p.dev[d].rows > k) flags |= IRLOSC_FLAG_BAD_JIDX;
        }
    }
    // fold the null-space term into the task vector: w = u_task_all [+ ext_f] - kvn * dx
    if (lane < k) wk[lane] -= kvn * dxs[lane];

    // ---- Cholesky of M, Y = L^-1 J^T, A = Y^T Y ------------------------------------------------------
    T detM;
    if (!wave_cholesky<T>(Ms, n, ldn, lane, &detM)) flags |= IRLOSC_FLAG_M_NOT_PD;
    {
        const int r = lane >> 2, q = lane & 3;
        for (int i = 0; i < n; ++i) {
            T s = T(0);
            if (r < k)
                for (int c = q; c < i; c += 4) s += Ms[i * ldn + c] * Ys[r * ldn + c];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            if (r < k && q == 0) Ys[r * ldn + i] = (Js[r * ldn + i] - s) / Ms[i * ldn + i];
            __syncthreads();
        }
    }
    for (int e = lane; e < k * k; e += 64) {
        int r = e / k, s = e % k;
        if (s <= r) {
            T a = T(0);
            for (int i = 0; i < n; ++i) a += Ys[r * ldn + i] * Ys[s * ldn + i];
            As[r * ldk + s] = a;
            As[s * ldk + r] = a;
            Ls[r * ldk + s] = a;
            Ls[s * ldk + r] = a;
        }
    }
    __syncthreads();

    // ---- t = Mx w ----------------------------------------------------------------------------------
    T detA;
    bool pd = wave_cholesky<T>(Ls, k, ldk, lane, &detA);
    // W = L_A^-1 (lower), lane j owns column j
    if (lane < k) {
        const int j = lane;
        for (int i = 0; i < k; ++i) {
            if (i < j) { Ws[i * ldk + j] = T(0); continue; }
            T s = (i == j) ? T(1) : T(0);
            for (int c = j; c < i; ++c) s -= Ls[i * ldk + c] * Ws[c * ldk + j];
            Ws[i * ldk + j] = s / Ls[i * ldk + i];
        }
    }
    __syncthreads();
    T nA2 = T(0), nW2 = T(0);
    for (int e = lane; e < k * k; e += 64) {
        T a = As[(e / k) * ldk + (e % k)];
        T w = Ws[(e / k) * ldk + (e % k)];
        nA2 += a * a;
        nW2 += w * w;
    }
    nA2 = wave_sum(nA2);
    nW2 = wave_sum(nW2);
    const bool small_det = !(t_abs(detA) >= T(1e-4));
    const T cond_bound = t_sqrt(nA2) * nW2;  // >= cond_2(A) when A is SPD
    const bool plain = pd && t_finite(cond_bound) && (!small_det || cond_bound < T(0.99e5));
    if (small_det) flags |= IRLOSC_FLAG_PINV_BRANCH;

    if (plain) {
        if (lane < k) {
            T s = T(0);
            for (int c = 0; c <= lane; ++c) s += Ws[lane * ldk + c] * wk[c];
            zk[lane] = s;
        }
        __syncthreads();
        if (lane < k) {
            T s = T(0);
            for (int i = lane; i < k; ++i) s += Ws[i * ldk + lane] * zk[i];
            tk[lane] = s;
        }
        __syncthreads();
    } else {
        // Cyclic two-sided Jacobi on A (symmetric k x k), eigenvectors accumulated in Ws = V.
        flags |= IRLOSC_FLAG_EIGEN_PATH;
        for (int e = lane; e < k * k; e += 64) Ws[(e / k) * ldk + (e % k)] = (e / k == e % k) ? T(1) : T(0);
        __syncthreads();
        const int max_sweeps = sizeof(T) == 8 ? 14 : 10;
        for (int sweep = 0; sweep < max_sweeps; ++sweep) {
            T off = T(0), dia = T(0);
            for (int e = lane; e < k * k; e += 64) {
                T a = As[(e / k) * ldk + (e % k)];
                if (e / k == e % k) dia += a * a; else off += a * a;
            }
            off = wave_sum(off);
            dia = wave_sum(dia);
            if (!(off > Eps<T>::v * Eps<T>::v * dia)) break;
            for (int pp = 0; pp < k - 1; ++pp) {
                for (int qq = pp + 1; qq < k; ++qq) {
                    T apq = As[pp * ldk + qq];
                    T app = As[pp * ldk + pp];
                    T aqq = As[qq * ldk + qq];
                    __syncthreads();
                    if (apq != T(0) && t_abs(apq) > Eps<T>::v * T(1e-3) * t_sqrt(t_abs(app * aqq))) {
                        T theta = (aqq - app) / (T(2) * apq);
                        T tt = (theta >= T(0) ? T(1) : T(-1)) / (t_abs(theta) + t_sqrt(theta * theta + T(1)));
                        T c = T(1) / t_sqrt(tt * tt + T(1));
                        T s = tt * c;
                        if (lane < k) {  // columns p,q
                            T aip = As[lane * ldk + pp], aiq = As[lane * ldk + qq];
                            As[lane * ldk + pp] = c * aip - s * aiq;
                            As[lane * ldk + qq] = s * aip + c * aiq;
                            T vip = Ws[lane * ldk + pp], viq = Ws[lane * ldk + qq];
                            Ws[lane * ldk + pp] = c * vip - s * viq;
                            Ws[lane * ldk + qq] = s * vip + c * viq;
                        }
                        __syncthreads();
                        if (lane < k) {  // rows p,q
                            T api = As[pp * ldk + lane], aqi = As[qq * ldk + lane];
                            As[pp * ldk + lane] = c * api - s * aqi;
                            As[qq * ldk + lane] = s * api + c * aqi;
                        }
                        __syncthreads();
                    }
                }
            }
        }
        __syncthreads();
        // eigenvalues on the diagonal; reference semantics on singular values |lambda|
        T lam = (lane < k) ? As[lane * ldk + lane] : T(0);
        T lmax = wave_max((lane < k) ? t_abs(lam) : T(0));
        // det from the spectrum (the Cholesky det is meaningless when A was not PD)
        T det = T(1);
        for (int i = 0; i < k; ++i) det *= As[i * ldk + i];
        const bool trunc = !(t_abs(det) >= T(1e-4));
        if (trunc) flags |= IRLOSC_FLAG_PINV_BRANCH; else flags &= ~IRLOSC_FLAG_PINV_BRANCH;
        bool cut = false;
        if (lane < k) {
            T s = T(0);
            for (int i = 0; i < k; ++i) s += Ws[i * ldk + lane] * wk[i];  // (V^T w)_lane
            cut = trunc && !(t_abs(lam) > T(1e-5) * lmax);
            zk[lane] = cut ? T(0) : s / lam;
        }
        if (__ballot(cut)) flags |= IRLOSC_FLAG_TRUNCATED;
        __syncthreads();
        if (lane < k) {
            T s = T(0);
            for (int i = 0; i < k; ++i) s += Ws[lane * ldk + i] * zk[i];
            tk[lane] = s;
        }
        __syncthreads();
    }

    // ---- joint torques ---------------------------------------------------------------------------------
    bool bad = false;
    if (lane < n) {
        T u = T(0);
        for (int d = 0; d < ndev; ++d) {  // branch A damping, osc.py:174 (assignment, device order)
            if (brA[d] && (p.dev[d].joint_mask & (1u << lane)))
                u = -T(gbase[d * IRLOSC_GAIN_WORDS + 1]) * mdq[lane];
        }
        T s = T(0);
        for (int r = 0; r < k; ++r) s += Js[r * ldn + lane] * tk[r];
        u -= s;
        if (p.cfgflags & IRLOSC_USE_G) u += p.bias[(size_t)b * n + lane];
        u -= kvn * mdq[lane];
        p.u[(size_t)b * n + lane] = u;
        bad = !t_finite(u);
    }
    if (__ballot(bad)) flags |= IRLOSC_FLAG_NONFINITE;
    if (lane == 0) p.flags[b] = flags;
    __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(64) void osc_generic_kernel(const KParams<T> p) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    // identity mode: block i handles instance b0 + i; worklist mode: grid-stride over index[0..*index_count)
    const int wl_count = p.index ? *p.index_count : 0;
    for (int it = blockIdx.x; p.index ? (it < wl_count) : (it == (int)blockIdx.x); it += gridDim.x)
        generic_instance<T>(p, p.index ? p.index[it] : p.b0 + it, smem);
}

template <typename T>
inline size_t generic_smem_bytes(int n, int k, int ndev) {
    int ldn = n | 1, ldk = k | 1;
    size_t words = (size_t)n * ldn + 2 * (size_t)k * ldn + 3 * (size_t)k * ldk + 2 * n + 4 * k;
    return words * sizeof(T) + ndev * sizeof(int) + 16;
}

}  // namespace irlosc
