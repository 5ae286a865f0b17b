// State assembly kernel: raw simulator arrays -> the resident C-ABI records of one slot.
//
// Batched counterpart of what the reference does per robot and per tick in Python:
//   M    = fullM.reshape(nv, nv)[np.ix_(ids, ids)]                                   robot.py:68-72
//   dq   = zeros(n); dq[dev.joint_ids_all] = qvel[dev.joint_ids_all]                 robot.py:60-65, device.py:90
//   bias = qfrc_bias[ids]                                                            osc.py:191
//   J_d  = vstack(jacp.reshape(3, nv), jacr.reshape(3, nv))[ctrlr_dof][:, ids]       device.py:123-132, robot.py:44-58
//   J    = vstack(J_d for d in targets)                                              osc.py:134-138
//   ee   = [xpos, xquat] of the EE body                                              device.py:97-99
//   F    = R(site) @ sensordata[f0:f0+3], tau = R(site) @ sensordata[t0:t0+3]        device.py:135-170
// Pure gathers plus one 3 x 3 rotation: HBM-bound, one 64-thread block per instance.
#pragma once
#include "osc_common.hpp"

namespace irlosc {

struct RawDesc {
    int32_t nv, n_sensor, n, k, ndev;
    int32_t joint_ids[IRLOSC_MAX_N];
    int32_t dq_src[IRLOSC_MAX_N];
    int32_t ft_force0[IRLOSC_MAX_DEV];
    int32_t ft_torque0[IRLOSC_MAX_DEV];
    uint32_t dofmask[IRLOSC_MAX_DEV];
    // qM as MuJoCo keeps it (irlosc_upload_raw_sparse): nM > 0 -- r.qM is then [B][nM], the run of raw dof i starts at madr[i] and walks
    // up the tree through par[] (mj_fullM's loop, robot.py:68-72); pos[j] = position of raw dof j in the robot's n-vector, -1: none
    int32_t nM;
    int16_t madr[IRLOSC_MAX_NV], par[IRLOSC_MAX_NV], pos[IRLOSC_MAX_NV];
};

template <typename T>
struct RawPtrs {
    const T* qM; const T* qvel; const T* qfrc_bias; const T* jacp; const T* jacr;
    const T* ee_xpos; const T* ee_xquat; const T* site_xmat; const T* sensordata;   // last two may be null
    T* M; T* J; T* dq; T* bias; T* ee; T* wrench;
};

template <typename T>
__global__ __launch_bounds__(64) void osc_assemble_kernel(const RawDesc d, const RawPtrs<T> r, const int B) {
    const int lane = threadIdx.x;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const int n = d.n, nv = d.nv, ndev = d.ndev;
        if (d.nM > 0) {
            // mj_fullM on the device: zeros, then row i's run up the tree, mirrored (lane = robot position; a pair (p, ancestor) is
            // written by lane p only)
            T* Mo = r.M + (size_t)b * n * n;
            for (int e = lane; e < n * n; e += 64) Mo[e] = T(0);
            __syncthreads();
            if (lane < n) {
                const T* q = r.qM + (size_t)b * d.nM;
                int adr = d.madr[d.joint_ids[lane]];
                for (int j = d.joint_ids[lane]; j >= 0; j = d.par[j], ++adr) {
                    const int pj = d.pos[j];
                    if (pj >= 0) { const T v = q[adr]; Mo[lane * n + pj] = v; Mo[pj * n + lane] = v; }
                }
            }
        } else {
            const T* qM = r.qM + (size_t)b * nv * nv;
            for (int e = lane; e < n * n; e += 64) {
                const int i = e / n, j = e - i * n;
                r.M[(size_t)b * n * n + e] = qM[(size_t)d.joint_ids[i] * nv + d.joint_ids[j]];
            }
        }
        if (lane < n) {
            const int src = d.dq_src[lane];
            r.dq[(size_t)b * n + lane] = src >= 0 ? r.qvel[(size_t)b * nv + src] : T(0);
            r.bias[(size_t)b * n + lane] = r.qfrc_bias[(size_t)b * nv + d.joint_ids[lane]];
        }
        int row = 0;
        for (int dv = 0; dv < ndev; ++dv) {
            const T* jp = r.jacp + ((size_t)b * ndev + dv) * 3 * nv;
            const T* jr = r.jacr + ((size_t)b * ndev + dv) * 3 * nv;
            for (int c = 0; c < 6; ++c) {
                if (!(d.dofmask[dv] & (1u << c))) continue;
                const T* src = (c < 3 ? jp + c * nv : jr + (c - 3) * nv);
                if (lane < n) r.J[((size_t)b * d.k + row) * n + lane] = src[d.joint_ids[lane]];
                ++row;
            }
            if (lane < 7) {
                const T v = lane < 3 ? r.ee_xpos[((size_t)b * ndev + dv) * 3 + lane]
                                     : r.ee_xquat[((size_t)b * ndev + dv) * 4 + (lane - 3)];
                r.ee[((size_t)b * ndev + dv) * 7 + lane] = v;
            } else if (lane >= 8 && lane < 14) {
                const int o = lane - 8, part = o / 3, i = o - part * 3;     // part 0: force, 1: torque
                const int s0 = part == 0 ? d.ft_force0[dv] : d.ft_torque0[dv];
                T v = T(0);
                if (s0 >= 0 && r.site_xmat && r.sensordata) {
                    const T* R = r.site_xmat + ((size_t)b * ndev + dv) * 9;
                    const T* s = r.sensordata + (size_t)b * d.n_sensor + s0;
                    // np.matmul(R, s): sum in index order, separate multiply and add like NumPy's dot of three terms
                    v = R[i * 3 + 0] * s[0];
                    v = v + R[i * 3 + 1] * s[1];
                    v = v + R[i * 3 + 2] * s[2];
                }
                r.wrench[((size_t)b * ndev + dv) * 6 + o] = v;
            }
        }
    }
}

// Symmetry probe for irlosc_upload / irlosc_tick on the throughput paths (which read row j of M as its column j): counts the
// instances with max |M - M^T| > 1e-6 max |M| in out[0] and remembers the first one as out[1] = max(INT_MAX - b), so that
// both words start from zero (one memset).  Only pairs of FINITE entries are judged: a robot whose M holds NaN / Inf (a
// diverged simulation) is not "asymmetric" -- it goes through to the kernel, which reports it per instance
// (IRLOSC_FLAG_NONFINITE / M_NOT_PD) while the other robots of the batch get their torques, like the reference, which
// simply propagates the NaN of that robot.  One 64-thread block per instance, grid-strided; one pass over M (0.16 ms per
// 65 536 instances, against ~10 ms of PCIe for the same records).
template <typename T>
__global__ __launch_bounds__(64) void osc_symmetry_kernel(const T* __restrict__ M, const int n, const int B, int32_t* __restrict__ out) {
    const int lane = threadIdx.x;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const T* Mb = M + (size_t)b * n * n;
        double asym = 0.0, scale = 0.0;
        for (int e = lane; e < n * n; e += 64) {
            const int i = e / n, j = e - i * n;
            const double v = (double)Mb[e], w = (double)Mb[j * n + i];
            if (t_finite(v) && t_finite(w)) {
                asym = fmax(asym, fabs(v - w));
                scale = fmax(scale, fabs(v));
            }
        }
        asym = wave_max(asym);
        scale = wave_max(scale);
        if (lane == 0 && asym > 1e-6 * fmax(scale, 1e-300)) {
            atomicAdd(&out[0], 1);
            atomicMax(&out[1], 0x7fffffff - b);
        }
    }
}

// Zero pattern of the records (dense-record form of the tree-structured factorisation, osc_row16.hpp): counts the instances
// with a non-zero entry of M outside mrow[j] or a non-zero column of J outside jcols.  Exact zeros are asked for: that is
// what mj_fullM / mj_jacBody (and the front end here) leave where the tree has no coupling.
template <typename T>
__global__ __launch_bounds__(64) void osc_structure_kernel(const T* __restrict__ M, const T* __restrict__ J, const int n, const int k,
                                                           const int B, const StructureMasks m, int32_t* __restrict__ out) {
    const int lane = threadIdx.x;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const T* Mb = M + (size_t)b * n * n;
        const T* Jb = J + (size_t)b * k * n;
        bool bad = false;
        const int c = lane & 31, half = lane >> 5;                 // two rows per pass, lane = column (n <= 32; no division)
        if (c < n) {
            for (int i = half; i < n; i += 2) bad = bad || (!((m.mrow[i] >> c) & 1u) && !(Mb[i * n + c] == (T)0));   // NaN counts as non-zero
            if (!((m.jcols >> c) & 1u))
                for (int r = half; r < k; r += 2) bad = bad || !(Jb[r * n + c] == (T)0);
        }
        if (__any(bad) && lane == 0) atomicAdd(&out[0], 1);
    }
}

}  // namespace irlosc
