"""Development aid: NumPy prototype of the LANE-PER-ROBOT OSC step (csrc/osc_lane.hpp) -- the same operation sequence, vectorised
over the batch axis instead of the 64 lanes of a wave -- checked against the oracle on records of random physical states.

  python tools/lane_proto.py [--layout k13] [--B 512] [--seed 3]

What it validates before any HIP is written: the right-looking tree factorisation in "accumulated update" form (Delta), the
canonical row order (rows grouped by end-effector body, not in targets order), padded rows, the certificate, the folded null-space
term.  Robots that fail the certificate are finished with numpy's pinv here (the HIP path hands them to the eigen pass).
Test infrastructure / tooling only: imports oracle/.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from irl_control_amd import synth  # noqa: E402
from irl_control_amd.rigid_body import DUAL_UR5_EE, RigidBodyModel  # noqa: E402
from oracle import osc_oracle, rigid_body as rb  # noqa: E402

# hinge tree of the Dual-UR5 (MuJoCo's depth-first numbering): parent hinge of every hinge
PARENT = [-1, 0, 1, 2, 3, 4, 5, 6, 7, 6, 6, 10, 6, 0, 13, 14, 15, 16, 17, 18, 19, 18, 18, 22, 18]
NJ = 25
# end-effector candidates: hinges that move them (stand dummy: hinge 0; right EE: 0..6; left EE: 0, 13..18)
CAND_HINGES = {"base": [0], "ur5right": [0, 1, 2, 3, 4, 5, 6], "ur5left": [0, 13, 14, 15, 16, 17, 18]}
CAND_ORDER = ["base", "ur5right", "ur5left"]


def ancestors(j):
    out = []
    p = PARENT[j]
    while p >= 0:
        out.append(p)
        p = PARENT[p]
    return sorted(out)


def lane_step(lay, gains, M, J, dq, bias, wrows, rows_cap=(1, 6, 6)):
    """wrows[B, k]: part 1 of the task signal in TARGETS order (what the task pass leaves).  Returns (u, plain, info)."""
    B = M.shape[0]
    names = lay.dev_names
    # ---- canonical rows: candidate by candidate, up to rows_cap[c] rows each; (external row, or -1 for padding) -------------
    ext_of = []          # canonical row -> external row (-1: padding)
    cand_of = []
    r0 = 0
    ext_row = {}
    for d, nm in enumerate(names):
        for i in range(6):
            if lay.ctrlr_dof[d][i]:
                ext_row.setdefault(nm, []).append(r0)
                r0 += 1
    for c, nm in enumerate(CAND_ORDER):
        have = ext_row.get(nm, [])
        assert len(have) <= rows_cap[c], "layout does not fit this instantiation"
        for s in range(rows_cap[c]):
            ext_of.append(have[s] if s < len(have) else -1)
            cand_of.append(nm)
    KC = len(ext_of)
    Jc = np.zeros((B, KC, NJ))
    wc = np.zeros((B, KC))
    for r, e in enumerate(ext_of):
        if e >= 0:
            Jc[:, r] = J[:, e]
            wc[:, r] = wrows[:, e]
    real_row = np.array([e >= 0 for e in ext_of])
    # ---- main recursion, hinges NJ-1 .. 0 ------------------------------------------------------------------------------------
    Delta = {}           # (i2, i1), i2 >= i1: accumulated sum of l_i2 l_i1 over the eliminated hinges below both
    DJ = np.zeros((B, KC, NJ))
    A = np.zeros((B, KC, KC))
    mdq = np.zeros((B, NJ))
    dx = np.zeros((B, KC))
    npd = np.zeros(B, dtype=bool)
    for j in range(NJ - 1, -1, -1):
        anc = ancestors(j)
        mjj = M[:, j, j]
        mdq[:, j] += mjj * dq[:, j]
        d = mjj - Delta.pop((j, j), 0.0)
        l = {}
        for i in anc:
            mji = M[:, j, i]
            mdq[:, j] += mji * dq[:, i]
            mdq[:, i] += mji * dq[:, j]
            l[i] = mji - Delta.pop((j, i), 0.0)
        npd |= ~(d > 0)
        d = np.maximum(d, 1e-300)
        rs = 1.0 / np.sqrt(d)
        for i in anc:
            l[i] = l[i] * rs
        for a in anc:
            for b2 in anc:
                if b2 <= a:
                    Delta[(a, b2)] = Delta.get((a, b2), 0.0) + l[a] * l[b2]
        rows_j = [r for r in range(KC) if j in CAND_HINGES[cand_of[r]]]
        y = {}
        for r in rows_j:
            dx[:, r] += Jc[:, r, j] * dq[:, j]
            y[r] = (Jc[:, r, j] - DJ[:, r, j]) * rs
            for i in anc:
                DJ[:, r, i] += l[i] * y[r]
        for r in rows_j:
            for s in rows_j:
                if s <= r:
                    A[:, r, s] += y[r] * y[s]
    assert not Delta, Delta.keys()
    iu = np.triu_indices(KC, 1)
    A[:, iu[0], iu[1]] = A[:, iu[1], iu[0]]
    # ---- w = task rows - kvn dx ----------------------------------------------------------------------------------------------
    kvn = np.broadcast_to(np.asarray(gains["null_kv"], dtype=np.float64), (B,)) if lay.nullspace else np.zeros(B)
    w = wc - kvn[:, None] * dx
    # ---- k x k: L D L^T with padded / exact-zero rows taken as pivot 1, certificate -------------------------------------------
    zr = real_row[None, :] & (np.einsum("bii->bi", A) == 0.0)
    notreal = (~real_row)[None, :] | zr
    w = np.where(notreal, 0.0, w)
    nA2 = (A * A).sum(axis=(1, 2))
    Lw = A.copy()
    dd = np.ones((B, KC))
    pd = np.ones(B, dtype=bool)
    det = np.ones(B)
    Lt = np.zeros((B, KC, KC))
    for j in range(KC):
        d = Lw[:, j, j].copy()
        bad = ~notreal[:, j] & ~(d > 0)
        pd &= ~bad
        d = np.where(bad | notreal[:, j], 1.0, d)
        det *= d
        dd[:, j] = d
        for i in range(j + 1, KC):
            Lt[:, i, j] = Lw[:, i, j] / d
        for i in range(j + 1, KC):
            for c in range(j + 1, i + 1):
                Lw[:, i, c] -= Lt[:, i, j] * Lw[:, c, j]
    # trace(A^-1) over the real rows: columns of W = L~^-1
    trA = np.zeros(B)
    for m in range(KC):
        x = np.zeros((B, KC))
        x[:, m] = 1.0
        for c in range(m + 1, KC):
            acc = np.zeros(B)
            for jj in range(m, c):
                acc -= Lt[:, c, jj] * x[:, jj]
            x[:, c] = acc
        contrib = (x * x / dd)
        contrib = np.where(notreal, 0.0, contrib)
        # column m of W only counts where row m itself is real (a padded row's unit column has one entry: its own, masked above)
        trA += contrib.sum(axis=1)
    small_det = ~pd | ~(np.abs(det) >= 1e-4) | zr.any(axis=1)
    cond_bound = np.sqrt(nA2) * trA
    plain = pd & np.isfinite(cond_bound) & (~small_det | (cond_bound < 0.99e5))
    # t = A^-1 w through the factor
    z = w.copy()
    for c in range(KC):
        for jj in range(c):
            z[:, c] -= Lt[:, c, jj] * z[:, jj]
    z = z / dd
    t = z.copy()
    for c in range(KC - 1, -1, -1):
        for i in range(c + 1, KC):
            t[:, c] -= Lt[:, i, c] * t[:, i]
    # the robots the certificate does not clear: truncated pseudo-inverse (the eigen pass of the HIP path)
    for b in np.flatnonzero(~plain):
        t[b] = np.linalg.pinv(A[b], rcond=1e-5) @ w[b]
    # ---- torques --------------------------------------------------------------------------------------------------------------
    u = np.zeros((B, NJ))
    kv = np.broadcast_to(np.asarray(gains["kv"], dtype=np.float64), (B, len(names)))
    for d, nm in enumerate(names):
        ids = np.asarray(lay.joint_ids[d])
        u[:, ids] = -kv[:, d:d + 1] * mdq[:, ids]
    if lay.use_g:
        u += bias
    u -= kvn[:, None] * mdq
    u -= np.einsum("bkn,bk->bn", Jc, t)
    return u, plain, dict(A=A, det=det, trA=trA, small_det=small_det, npd=npd)


def task_rows(lay, gains, ee, tgt, wrench=None):
    """Part 1 of the task signal (+ the wrench): what the task pass leaves, in targets order (through the oracle's own helpers)."""
    B = ee.shape[0]
    out = np.zeros((B, lay.k))
    g = {k2: np.asarray(v, dtype=np.float64) for k2, v in gains.items()}
    for b in range(B):
        r = 0
        for d in range(lay.ndev):
            dof = np.asarray(lay.ctrlr_dof[d], dtype=bool)
            e = osc_oracle.calc_error(ee[b, d, :3], ee[b, d, 3:], tgt[b, d, :3], tgt[b, d, 3:], dof[:3], dof[3:])
            kp, kv, ko = g["kp"][d], g["kv"][d], g["ko"][d]
            e = osc_oracle.limit_vel(e, g["max_vel"][d], kp, kv, ko) * np.array(list(g["k"][d]) + [1.0] * 3)
            if wrench is not None:
                e = e + wrench[b, d]
            n = int(dof.sum())
            out[b, r:r + n] = e[dof]
            r += n
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layout", default="k13")
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--seed", type=int, default=3)
    a = ap.parse_args()
    lay = synth.make_layout(a.layout)
    model = RigidBodyModel.load("dual_ur5")
    rng = np.random.default_rng(a.seed)
    qpos, qvel = model.random_state(rng, a.B)
    om = rb.Model()
    ld = lay.as_oracle_dict()
    ee_names = {nm: DUAL_UR5_EE[nm] for nm in lay.dev_names}
    recs = [rb.records(om, ld, ee_names if False else DUAL_UR5_EE, qpos[b], qvel[b]) for b in range(a.B)]
    st = {k2: np.array([r[k2] for r in recs]) for k2 in ("M", "J", "dq", "bias", "ee_pose")}
    _, gains, g = synth.make_batch(a.layout, a.B, seed=a.seed + 1)
    tgt = synth.targets_near(st["ee_pose"], rng)
    wrench = g.get("wrench") if lay.admittance else None
    ref = osc_oracle.generate_batch(ld, gains, st["M"], st["J"], st["dq"], st["bias"], st["ee_pose"], tgt, wrench)
    wr = task_rows(lay, gains, st["ee_pose"], tgt, wrench)
    cap = (1, 6, 6)
    u, plain, info = lane_step(lay, gains, st["M"], st["J"], st["dq"], st["bias"], wr, cap)
    rel = np.max(np.abs(u - ref), axis=1) / np.max(np.abs(ref), axis=1)
    print(f"layout {a.layout}: {a.B} robots, plain {plain.mean():.3f}, small_det {info['small_det'].mean():.3f}, "
          f"max rel err plain {rel[plain].max():.3e}, flagged {rel[~plain].max() if (~plain).any() else 0:.3e}")


if __name__ == "__main__":
    main()
