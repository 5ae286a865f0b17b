"""ORACLE — test infrastructure only.  CPU restatement (float64 NumPy, op for op) of the reference's
operational-space-control hot path, /root/reference/irl_control/osc.py:41-210.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (irl_control_amd/) never does.  It works on plain arrays so it can travel to the GPU box
(the reference itself cannot).

Pinning: the reference ships no golden vectors (its test-suite is ``assert True``,
/root/reference/irl_control/tests/run_tests.py:1), so this restatement is pinned against OUTPUTS OF
THE REFERENCE ITSELF, imported in the build container by oracle/make_golden.py (fixtures under
tests/golden/, checked by tests/test_oracle_golden.py to <=1e-12).  One dependency of the reference
is absent everywhere (transforms3d, unpinned in /root/reference/requirements.in:3): its sxyz
quaternion/Euler formulas are restated in ``_quat2euler_sxyz`` etc. below and cross-checked against
scipy — for that part PARITY IS UNPINNED w.r.t. transforms3d itself.

Conventions: quaternions w,x,y,z; J rows stacked in *targets order*; joint-space vectors have
length n = robot.num_joints_total.
"""
import math
from typing import Dict, List, Optional, Sequence

import numpy as np

_EPS = float(np.finfo(np.float64).eps)


# --------------------------------------------------------------------------------------------
# transforms3d pieces (call sites osc.py:115-117) — restated formulas, axes 'sxyz'
# --------------------------------------------------------------------------------------------
def _qmult(q1, q2):
    w1, x1, y1, z1 = q1
    w2, x2, y2, z2 = q2
    return (w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
            w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
            w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
            w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2)


def _qconj(q):
    q = np.asarray(q, dtype=np.float64)
    return np.array([q[0], -q[1], -q[2], -q[3]])


def _quat2euler_sxyz(q):
    w, x, y, z = q
    Nq = w * w + x * x + y * y + z * z
    if Nq < _EPS:
        R = np.eye(3)
    else:
        s = 2.0 / Nq
        X, Y, Z = x * s, y * s, z * s
        wX, wY, wZ = w * X, w * Y, w * Z
        xX, xY, xZ = x * X, x * Y, x * Z
        yY, yZ, zZ = y * Y, y * Z, z * Z
        R = np.array([[1.0 - (yY + zZ), xY - wZ, xZ + wY],
                      [xY + wZ, 1.0 - (xX + zZ), yZ - wX],
                      [xZ - wY, yZ + wX, 1.0 - (xX + yY)]])
    cy = math.sqrt(R[0, 0] * R[0, 0] + R[1, 0] * R[1, 0])
    if cy > 4.0 * _EPS:
        return (math.atan2(R[2, 1], R[2, 2]), math.atan2(-R[2, 0], cy), math.atan2(R[1, 0], R[0, 0]))
    return (math.atan2(-R[1, 2], R[1, 1]), math.atan2(-R[2, 0], cy), 0.0)


# --------------------------------------------------------------------------------------------
# osc.py helpers
# --------------------------------------------------------------------------------------------
def svd_inverse(A):
    """osc.py:59-68 (__svd_solve): V diag(1/s) U^T, no truncation."""
    u, s, v = np.linalg.svd(A)
    return np.dot(v.transpose(), np.dot(np.diag(s ** -1), u.transpose()))


def task_inertia(J, M):
    """osc.py:41-56 (__Mx) -> (Mx, M_inv, Mx_inv, det)."""
    M_inv = svd_inverse(M)
    Mx_inv = np.dot(J, np.dot(M_inv, J.T))
    threshold = 1e-4
    det = np.linalg.det(Mx_inv)
    if abs(det) >= threshold:
        Mx = svd_inverse(Mx_inv)
    else:
        Mx = np.linalg.pinv(Mx_inv, rcond=threshold * 0.1)
    return Mx, M_inv, Mx_inv, det


def limit_vel(u_task, max_vel, kp, kv, ko):
    """osc.py:70-99 (__limit_vel); lamb = [kp]*3+[ko]*3 / kv (osc.py:35-39)."""
    lamb = np.array([kp] * 3 + [ko] * 3) / kv
    scale = np.ones(6)
    norm_xyz = np.linalg.norm(u_task[:3])
    sat_xyz = max_vel[0] / kp * kv
    if norm_xyz > sat_xyz:
        scale[:3] *= sat_xyz / norm_xyz
    norm_abg = np.linalg.norm(u_task[3:])
    sat_abg = max_vel[1] / ko * kv
    if norm_abg > sat_abg:
        scale[3:] *= sat_abg / norm_abg
    return kv * scale * lamb * u_task


def calc_error(ee_xyz, ee_quat, tgt_xyz, tgt_quat, dof_xyz, dof_abg):
    """osc.py:101-118 (calc_error)."""
    u_task = np.zeros(6)
    if np.sum(dof_xyz) > 0:
        u_task[:3] = np.asarray(ee_xyz) - np.asarray(tgt_xyz)
    if np.sum(dof_abg) > 0:
        t = np.asarray(tgt_quat, dtype=np.float64)
        q_d = t / math.sqrt(float((t ** 2).sum()))
        q_r = np.array(_qmult(q_d, _qconj(ee_quat)))
        u_task[3:] = _quat2euler_sxyz(_qconj(q_r))
    return u_task


def generate(M, J_list: Sequence[np.ndarray], dq, bias, devs: List[Dict], null_kv: Optional[float],
             use_g: bool = True, admittance: bool = False, want_intermediates: bool = False):
    """osc.py:120-200 for ONE robot instance, from already-assembled inputs.

    J_list[i] : [r_i, n] masked Jacobian of target device i (targets order)  (osc.py:134-138)
    devs[i]   : dict(ee_xyz, ee_quat, tgt_xyz, tgt_quat, tgt_vel[6], wrench[6], ctrlr_dof[6] bool,
                     joint_ids_all (positions in the n-vector), J_idx (rows into dx, robot.py:50-55),
                     max_vel [2] or None, kp, kv, ko, k[3], d[3])
    returns u_all[n]  (+ dict of intermediates)
    """
    J = np.vstack(list(J_list))
    Mx, M_inv, Mx_inv, det = task_inertia(J, M)
    dx = np.dot(J, dq)
    uv_all = np.dot(M, dq)
    n = M.shape[0]
    u_all = np.zeros(n)
    u_task_all = np.array([])
    ext_f = np.array([])
    u_tasks = []
    for dv in devs:
        dof = np.asarray(dv["ctrlr_dof"], dtype=bool)
        # calc_error reads the LIVE device.ctrlr_dof_xyz / ctrlr_dof_abg (osc.py:108,113), the row mask below is the
        # device.ctrlr_dof frozen in Device.__init__ (device.py:36): they differ once a caller re-masks a live device
        # (examples/ps_move_example.py:137-150); "calc_xyz" / "calc_abg" carry the live ones when they do
        u_task = calc_error(dv["ee_xyz"], dv["ee_quat"], dv["tgt_xyz"], dv["tgt_quat"],
                            dv.get("calc_xyz", dof[:3]), dv.get("calc_abg", dof[3:]))
        stiffness = np.array(list(dv["k"]) + [1] * 3)
        damping = np.array(list(dv["d"]) + [1] * 3)
        kp, kv, ko = dv["kp"], dv["kv"], dv["ko"]
        if dv["max_vel"] is not None:
            u_task = limit_vel(u_task, dv["max_vel"], kp, kv, ko)
            u_task *= stiffness
        else:
            u_task *= np.array([kp] * 3 + [ko] * 3) * stiffness
        target_vel = np.asarray(dv["tgt_vel"], dtype=np.float64)
        if np.all(target_vel) == 0:                                   # osc.py:173 (sic)
            ids = np.asarray(dv["joint_ids_all"])
            u_all[ids] = -1 * kv * uv_all[ids]
        else:
            diff = dx[np.asarray(dv["J_idx"])] - target_vel[dof]
            u_task[dof] += kv * diff * damping[dof]
        force = np.asarray(dv["wrench"], dtype=np.float64)
        ext_f = np.append(ext_f, force[dof])
        u_task_all = np.append(u_task_all, u_task[dof])
        u_tasks.append(u_task.copy())
    if admittance:
        u_all -= np.dot(J.T, np.dot(Mx, u_task_all + ext_f))
    else:
        u_all -= np.dot(J.T, np.dot(Mx, u_task_all))
    u_after_task = u_all.copy()
    if use_g:
        u_all += bias
    if null_kv is not None:
        u_null = np.dot(M, -null_kv * dq)
        Jbar = np.dot(M_inv, np.dot(J.T, Mx))
        null_filter = np.eye(n) - np.dot(J.T, Jbar.T)
        u_all += np.dot(null_filter, u_null)
    if want_intermediates:
        return u_all, dict(Mx=Mx, M_inv=M_inv, Mx_inv=Mx_inv, det=det, u_tasks=np.array(u_tasks),
                           u_task_all=u_task_all, ext_f=ext_f, u_after_task=u_after_task)
    return u_all


# --------------------------------------------------------------------------------------------
# Batched driver over the C-ABI record layout (include/irlosc.h)
# --------------------------------------------------------------------------------------------
def generate_batch(layout: Dict, gains: Dict, M, J, dq, bias, ee_pose, tgt_pose,
                   wrench=None, tgt_vel=None, idx=None):
    """Loop ``generate`` over a batch stored in the C-ABI layout.

    layout : dict(n, dev_rows[ndev], ctrlr_dof[ndev][6], joint_ids[ndev] (lists of positions),
                  j_idx0[ndev] (first dx row used by branch B), use_g, admittance, nullspace
                  [, calc_xyz[ndev], calc_abg[ndev]: the live masks calc_error reads, default from ctrlr_dof])
    gains  : dict(kp, kv, ko [ndev]; k, d [ndev,3]; max_vel [ndev,2]; has_max_vel [ndev];
                  null_kv) — each either broadcast or with a leading batch axis.
    M[B,n,n] J[B,k,n] dq[B,n] bias[B,n] ee_pose[B,ndev,7] tgt_pose[B,ndev,7] wrench[B,ndev,6]
    tgt_vel[B,ndev,6].   Returns u[B,n] float64.
    """
    M = np.asarray(M, dtype=np.float64)
    B, n = M.shape[0], layout["n"]
    ndev = len(layout["dev_rows"])
    rows = np.concatenate([[0], np.cumsum(layout["dev_rows"])]).astype(int)
    out = np.zeros((B, n))
    sel = range(B) if idx is None else idx

    def g(name, b, d=None):
        a = np.asarray(gains[name], dtype=np.float64)
        full = {"kp": 1, "kv": 1, "ko": 1, "k": 2, "d": 2, "max_vel": 2, "null_kv": 0}[name]
        if a.ndim == full + 1:
            a = a[b]
        return a if d is None else a[d]

    for b in sel:
        devs = []
        for d in range(ndev):
            dof = np.asarray(layout["ctrlr_dof"][d], dtype=bool)
            r = int(layout["dev_rows"][d])
            devs.append(dict(
                ee_xyz=ee_pose[b, d, :3], ee_quat=ee_pose[b, d, 3:],
                tgt_xyz=tgt_pose[b, d, :3], tgt_quat=tgt_pose[b, d, 3:],
                tgt_vel=np.zeros(6) if tgt_vel is None else tgt_vel[b, d],
                wrench=np.zeros(6) if wrench is None else wrench[b, d],
                ctrlr_dof=dof, joint_ids_all=layout["joint_ids"][d],
                calc_xyz=(layout["calc_xyz"][d] if "calc_xyz" in layout else dof[:3]),
                calc_abg=(layout["calc_abg"][d] if "calc_abg" in layout else dof[3:]),
                J_idx=np.arange(layout["j_idx0"][d], layout["j_idx0"][d] + r),
                max_vel=(g("max_vel", b, d) if layout.get("has_max_vel", [True] * ndev)[d] else None),
                kp=float(g("kp", b, d)), kv=float(g("kv", b, d)), ko=float(g("ko", b, d)),
                k=g("k", b, d), d=g("d", b, d)))
        Jl = [np.asarray(J[b, rows[d]:rows[d + 1]], dtype=np.float64) for d in range(ndev)]
        nk = float(g("null_kv", b)) if layout["nullspace"] else None
        out[b] = generate(M[b], Jl, np.asarray(dq[b], dtype=np.float64),
                          np.asarray(bias[b], dtype=np.float64), devs, nk,
                          use_g=layout["use_g"], admittance=layout["admittance"])
    return out
