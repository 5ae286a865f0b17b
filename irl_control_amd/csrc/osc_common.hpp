// Shared device-side pieces of the OSC kernels (gfx950 / CDNA4 only).
//
// Everything here restates arithmetic of /root/reference/irl_control/osc.py; the line references
// in the comments are to that file unless another file is named.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/irlosc.h"

namespace irlosc {

// Copy of a plain struct that may live in the constant address space (train tables are read through it so that
// the loads are scalar): C++ will not bind an implicit copy constructor across address spaces.
template <typename T>
__device__ __forceinline__ T pod_copy(const T& src) { return src; }
template <typename T>
__device__ __forceinline__ T pod_copy(const __attribute__((address_space(4))) T& src) {
    static_assert(sizeof(T) % 4 == 0, "word-wise copy");
    T dst;
    const __attribute__((address_space(4))) uint32_t* w = (const __attribute__((address_space(4))) uint32_t*)&src;
    uint32_t* d = reinterpret_cast<uint32_t*>(&dst);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) d[i] = w[i];
    return dst;
}

// Per-target-device metadata, broadcast to every instance (lives in kernarg/SGPR space).
struct DevMeta {
    int32_t row0;         // first row of this device's block in the stacked J / task vector
    int32_t rows;         // r_d
    uint32_t dofmask;     // bits 0..5 = ctrlr_dof (device.py:36)
    uint32_t calc;        // bit0: xyz error computed (osc.py:108), bit1: abg error computed (osc.py:113)
    uint32_t joint_mask;  // bit j: position j belongs to device.joint_ids_all (osc.py:174)
    int32_t jidx0;        // first dx row used by the target-velocity branch (robot.py:50-55)
};

template <typename T>
struct KParams {
    const T* M;
    const T* J;
    const T* dq;
    const T* bias;
    const T* ee;      // [B][ndev][7]
    const T* tgt;     // [B][ndev][7]
    const T* tvel;    // [B][ndev][6] or nullptr
    const T* wrench;  // [B][ndev][6] or nullptr
    T* u;             // [B][n]
    uint32_t* flags;  // [B]
    const T* gains;   // [nb][ndev][12]
    const T* null_kv; // [nb]
    const int32_t* index;        // optional worklist of instance ids (nullptr = identity)
    const int32_t* index_count;  // device-side length of the worklist (grid-strided) when index != nullptr
    int32_t b0;                  // first instance handled by block 0 when index == nullptr
    unsigned long long* dbg;     // phase-timing buffer (IRLOSC_PHASE_TIMING=1 debug runs), else nullptr
    int32_t gains_per_instance;
    int32_t B, n, k, ndev;
    uint32_t cfgflags;
    int32_t padded;              // host-side launch choice (irlosc_create: IRLOSC_CLASS_ROW16_PADDED): the KMAX-padded row16 variant; kernels ignore it
    DevMeta dev[IRLOSC_MAX_DEV];
};

// ---- scalar math wrappers -----------------------------------------------------------------------
__device__ __forceinline__ float t_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double t_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float t_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double t_abs(double x) { return fabs(x); }
__device__ __forceinline__ float t_atan2(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ double t_atan2(double y, double x) { return atan2(y, x); }
template <typename T> struct Eps;
template <> struct Eps<float> { static constexpr float v = 1.1920929e-7f; };
template <> struct Eps<double> { static constexpr double v = 2.220446049250313e-16; };

__device__ __forceinline__ bool t_finite(float x) { return fabsf(x) <= 3.4028235e38f; }
__device__ __forceinline__ bool t_finite(double x) { return fabs(x) <= 1.7976931348623157e308; }

// ---- task-space error of one device (calc_error, osc.py:101-118) ----------------------------------
// ee/tg: x y z qw qx qy qz.  The quaternion part restates the transforms3d calls made there:
// q_d = normalized_vector(tgt_quat); q_r = qmult(q_d, qconjugate(q_ee)); e[3:] = quat2euler(qconjugate(q_r))
// with quat2euler(axes='sxyz') = mat2euler(quat2mat(q)).
template <typename T>
__device__ __forceinline__ void task_error6(const T* __restrict__ ee, const T* __restrict__ tg,
                                            bool cx, bool ca, T e[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) e[i] = T(0);
    if (cx) {
        e[0] = ee[0] - tg[0];
        e[1] = ee[1] - tg[1];
        e[2] = ee[2] - tg[2];
    }
    if (ca) {
        T tw = tg[3], tx = tg[4], ty = tg[5], tz = tg[6];
        T nrm = t_sqrt(tw * tw + tx * tx + ty * ty + tz * tz);
        T w1 = tw / nrm, x1 = tx / nrm, y1 = ty / nrm, z1 = tz / nrm;  // q_d
        T w2 = ee[3], x2 = -ee[4], y2 = -ee[5], z2 = -ee[6];            // conj(q_ee)
        // Hamilton product q_r = q_d * conj(q_ee)
        T rw = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2;
        T rx = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2;
        T ry = w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2;
        T rz = w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2;
        // q = conj(q_r)
        T w = rw, x = -rx, y = -ry, z = -rz;
        T Nq = w * w + x * x + y * y + z * z;
        T r00 = T(1), r10 = T(0), r20 = T(0), r21 = T(0), r22 = T(1), r11 = T(1), r12 = T(0);
        if (!(Nq < T(2.220446049250313e-16))) {
            T s = T(2) / Nq;
            T X = x * s, Y = y * s, Z = z * s;
            T wX = w * X, wY = w * Y, wZ = w * Z;
            T xX = x * X, xY = x * Y, xZ = x * Z;
            T yY = y * Y, yZ = y * Z, zZ = z * Z;
            r00 = T(1) - (yY + zZ);
            r10 = xY + wZ;
            r20 = xZ - wY;
            r21 = yZ + wX;
            r22 = T(1) - (xX + yY);
            r11 = T(1) - (xX + zZ);
            r12 = yZ - wX;
        }
        T cy = t_sqrt(r00 * r00 + r10 * r10);
        if (cy > T(4.0 * 2.220446049250313e-16)) {
            e[3] = t_atan2(r21, r22);
            e[4] = t_atan2(-r20, cy);
            e[5] = t_atan2(r10, r00);
        } else {
            e[3] = t_atan2(-r12, r11);
            e[4] = t_atan2(-r20, cy);
            e[5] = T(0);
        }
    }
}

// Gains record: kp kv ko k0 k1 k2 d0 d1 d2 max_vel0 max_vel1 has_max_vel
// Velocity limiting + gains + stiffness (osc.py:70-99 and 160-168).  e is updated in place.
template <typename T>
__device__ __forceinline__ void apply_gains6(const T* __restrict__ g, T e[6]) {
    const T kp = g[0], kv = g[1], ko = g[2];
    if (g[11] != T(0)) {
        T sx = T(1), sa = T(1);
        T nx = t_sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        T satx = g[9] / kp * kv;
        if (nx > satx) sx = sx * (satx / nx);
        T na = t_sqrt(e[3] * e[3] + e[4] * e[4] + e[5] * e[5]);
        T sata = g[10] / ko * kv;
        if (na > sata) sa = sa * (sata / na);
        T lp = kp / kv, lo = ko / kv;  // lamb (osc.py:39)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            e[i] = kv * sx * lp * e[i];
            e[3 + i] = kv * sa * lo * e[3 + i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) e[i] *= g[3 + i];  // stiffness k; abg entries are 1
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            e[i] *= kp * g[3 + i];
            e[3 + i] *= ko;
        }
    }
}

// Wave-wide sum over 64 lanes (result in every lane).
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        T o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// zero pattern asked of dense records (osc_structure_kernel): bit c of mrow[j]: M[j][c] may be non-zero; bit c of jcols: column c of J
struct StructureMasks { uint32_t mrow[32]; uint32_t jcols; };

}  // namespace irlosc
