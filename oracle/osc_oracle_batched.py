"""ORACLE, vectorised form — test infrastructure only (same rules as osc_oracle.py: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

The same arithmetic as oracle/osc_oracle.py (= /root/reference/irl_control/osc.py:41-200), but with the
batch axis carried through NumPy's stacked LAPACK calls (np.linalg.svd / det / pinv on [B, m, m]) instead of
a Python loop.  It exists to give bench.py a CPU baseline that is not dominated by interpreter overhead; it is
held to the loop oracle (which is the one pinned against the reference's own outputs) by
tests/test_oracle_golden.py::test_batched_oracle_matches_loop_oracle.
"""
from typing import Dict

import numpy as np

_EPS = float(np.finfo(np.float64).eps)


def _svd_inverse(A):
    """osc.py:59-68 on a stack of matrices: V diag(1/s) U^T, no truncation."""
    u, s, vt = np.linalg.svd(A)
    return np.matmul(np.swapaxes(vt, -1, -2) / s[..., None, :], np.swapaxes(u, -1, -2))


def _qmult(a, b):
    w1, x1, y1, z1 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    w2, x2, y2, z2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                     w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
                     w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2], axis=-1)


def _qconj(q):
    return q * np.array([1.0, -1.0, -1.0, -1.0])


def _quat2euler_sxyz(q):
    """transforms3d quat2mat + mat2euler('sxyz') (restated; see osc_oracle._quat2euler_sxyz)."""
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    Nq = w * w + x * x + y * y + z * z
    ok = Nq >= _EPS
    s = np.where(ok, 2.0 / np.where(ok, Nq, 1.0), 0.0)
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ = w * X, w * Y, w * Z
    xX, xY, xZ = x * X, x * Y, x * Z
    yY, yZ, zZ = y * Y, y * Z, z * Z
    R00, R10, R20 = 1.0 - (yY + zZ), xY + wZ, xZ - wY
    R11, R12, R21, R22 = 1.0 - (xX + zZ), yZ - wX, yZ + wX, 1.0 - (xX + yY)
    cy = np.sqrt(R00 * R00 + R10 * R10)
    reg = cy > 4.0 * _EPS
    ax = np.where(reg, np.arctan2(R21, R22), np.arctan2(-R12, R11))
    ay = np.arctan2(-R20, cy)
    az = np.where(reg, np.arctan2(R10, R00), 0.0)
    return np.stack([ax, ay, az], axis=-1)


def generate_batch(layout: Dict, gains: Dict, M, J, dq, bias, ee_pose, tgt_pose, wrench=None, tgt_vel=None):
    """Same contract as osc_oracle.generate_batch (C-ABI record layout in, u[B, n] float64 out)."""
    M = np.asarray(M, dtype=np.float64)
    J = np.asarray(J, dtype=np.float64)
    dq = np.asarray(dq, dtype=np.float64)
    B, n = M.shape[0], layout["n"]
    ndev = len(layout["dev_rows"])
    rows = np.concatenate([[0], np.cumsum(layout["dev_rows"])]).astype(int)

    def g(name, d=None):
        a = np.asarray(gains[name], dtype=np.float64)
        full = {"kp": 1, "kv": 1, "ko": 1, "k": 2, "d": 2, "max_vel": 2, "null_kv": 0}[name]
        if a.ndim == full:                                   # broadcast set -> leading batch axis
            a = np.broadcast_to(a, (B,) + a.shape)
        return a if d is None else a[:, d]

    # osc.py:41-56
    M_inv = _svd_inverse(M)
    Jt = np.swapaxes(J, -1, -2)
    Mx_inv = np.matmul(J, np.matmul(M_inv, Jt))
    det = np.linalg.det(Mx_inv)
    Mx = np.linalg.pinv(Mx_inv, rcond=1e-5)
    big = np.abs(det) >= 1e-4
    if np.any(big):
        Mx[big] = _svd_inverse(Mx_inv[big])
    dx = np.einsum("bkn,bn->bk", J, dq)
    uv_all = np.einsum("bij,bj->bi", M, dq)
    u_all = np.zeros((B, n))
    k = int(rows[-1])
    u_task_all = np.zeros((B, k))
    ext_f = np.zeros((B, k))
    has_mv = layout.get("has_max_vel", [True] * ndev)
    for d in range(ndev):
        dof = np.asarray(layout["ctrlr_dof"][d], dtype=bool)
        ee = np.asarray(ee_pose[:, d], dtype=np.float64)
        tg = np.asarray(tgt_pose[:, d], dtype=np.float64)
        u_task = np.zeros((B, 6))
        if dof[:3].sum() > 0:
            u_task[:, :3] = ee[:, :3] - tg[:, :3]
        if dof[3:].sum() > 0:
            q_d = tg[:, 3:] / np.sqrt((tg[:, 3:] ** 2).sum(axis=1, keepdims=True))
            q_r = _qmult(q_d, _qconj(ee[:, 3:]))
            u_task[:, 3:] = _quat2euler_sxyz(_qconj(q_r))
        kp, kv, ko = g("kp", d), g("kv", d), g("ko", d)
        gain6 = np.concatenate([np.repeat(kp[:, None], 3, 1), np.repeat(ko[:, None], 3, 1)], axis=1)
        stiff = np.concatenate([g("k", d), np.ones((B, 3))], axis=1)
        damp = np.concatenate([g("d", d), np.ones((B, 3))], axis=1)
        if has_mv[d]:
            mv = g("max_vel", d)
            scale = np.ones((B, 6))
            nx = np.linalg.norm(u_task[:, :3], axis=1)
            sx = mv[:, 0] / kp * kv
            m = nx > sx
            scale[m, :3] *= (sx[m] / nx[m])[:, None]
            na = np.linalg.norm(u_task[:, 3:], axis=1)
            sa = mv[:, 1] / ko * kv
            m = na > sa
            scale[m, 3:] *= (sa[m] / na[m])[:, None]
            u_task = kv[:, None] * scale * (gain6 / kv[:, None]) * u_task * stiff
        else:
            u_task = u_task * gain6 * stiff
        tv = np.zeros((B, 6)) if tgt_vel is None else np.asarray(tgt_vel[:, d], dtype=np.float64)
        branch_a = np.all(tv != 0.0, axis=1) == 0                         # osc.py:173 (sic): any zero component
        ids = np.asarray(layout["joint_ids"][d])
        u_all[np.ix_(branch_a, ids)] = -kv[branch_a, None] * uv_all[np.ix_(branch_a, ids)]
        if not np.all(branch_a):
            nb = ~branch_a
            r = int(layout["dev_rows"][d])
            j0 = layout["j_idx0"][d]
            diff = dx[nb][:, j0:j0 + r] - tv[nb][:, dof]
            sel = np.flatnonzero(dof)
            u_task[np.ix_(nb, sel)] += kv[nb, None] * diff * damp[nb][:, dof]
        if wrench is not None:
            ext_f[:, rows[d]:rows[d + 1]] = np.asarray(wrench[:, d], dtype=np.float64)[:, dof]
        u_task_all[:, rows[d]:rows[d + 1]] = u_task[:, dof]
    w = u_task_all + ext_f if layout["admittance"] else u_task_all
    u_all -= np.einsum("bkn,bk->bn", J, np.einsum("bij,bj->bi", Mx, w))
    if layout["use_g"]:
        u_all += np.asarray(bias, dtype=np.float64)
    if layout["nullspace"]:
        u_null = np.einsum("bij,bj->bi", M, -g("null_kv")[:, None] * dq)
        Jbar = np.matmul(M_inv, np.matmul(Jt, Mx))                        # [B, n, k]
        null_filter = np.eye(n)[None] - np.matmul(Jt, np.swapaxes(Jbar, -1, -2))
        u_all += np.einsum("bij,bj->bi", null_filter, u_null)
    return u_all
