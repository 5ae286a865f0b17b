#!/usr/bin/env python3
"""Wide parity sweep on the GPU: the row16 throughput kernel against the generic kernel (exact cyclic-Jacobi spectrum, itself
held to the reference's outputs by the goldens) on many synthetic batches; every disagreement over 1e-5 is then classified
with the float64 oracle (is a singular value of J M^-1 J^T within 1 % of the pinv cut, i.e. outside the parity domain?).
    python tools/parity_sweep.py [--seeds 16] [--batch 65536] [--layout k13] [--mode f64|mixed]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irl_control_amd import BatchedOSC, _lib, synth          # noqa: E402
from oracle import osc_oracle                                # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=16)
ap.add_argument("--seed0", type=int, default=0, help="first seed index (a later run continues where an earlier one stopped)")
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--layout", default="k13")
ap.add_argument("--mode", default="f64", choices=["f64", "mixed"], help="mixed: float32 records on the fp64-arithmetic row16 kernel")
ap.add_argument("--tol", type=float, default=1e-5)
ap.add_argument("--band", type=float, default=1e-2, help="a singular value this close to the cut (relative) puts an instance outside the parity domain")
ap.add_argument("--per-instance-gains", action="store_true")
ap.add_argument("--physical", action="store_true", help="M, J, bias, EE poses of random joint states of the Dual-UR5 model (rigid-body "
                "front end on the GPU) instead of the synthetic records: the conditioning of real kinematics, singular poses included")
ap.add_argument("--stress-rows", type=int, default=3, help="with --stress: up to this many rows (more than 3 under the cut sends an instance "
                "to the give-up list, i.e. through the generic kernel)")
ap.add_argument("--stress", action="store_true", help="scale 1-3 task rows of J per instance by 10^U(-3.5, -1.5): eigenvalues of "
                "J M^-1 J^T spread all over the neighbourhood of the pinv cut, up to three of them under it")
a = ap.parse_args()
dt = np.float64 if a.mode == "f64" else np.float32
B = a.batch
tot = bad_in = bad_out = 0
for sd in range(a.seed0, a.seed0 + a.seeds):
    lay, gains, g = synth.make_batch(a.layout, B, seed=777000 + 131 * sd, dtype=dt, per_instance_gains=a.per_instance_gains)
    if a.physical:
        from irl_control_amd.rigid_body import RigidBodyModel
        rngp = np.random.default_rng(9090 + sd)
        model = RigidBodyModel.load("dual_ur5")
        q = rngp.uniform(-np.pi, np.pi, (B, lay.n))
        q[:, [7, 8, 9, 10, 11, 12, 19, 20, 21, 22, 23, 24]] = rngp.uniform(0.0, 0.8, (B, 12))       # gripper joints
        q[rngp.random(B) < 0.1, 3] = 0.0                                   # a tenth with the right elbow stretched: singular arm
        qd = rngp.normal(0.0, 0.5, (B, lay.n))
        fe = BatchedOSC(lay, B, dtype=np.float64)
        fe.set_model(model)
        fe.upload_q(q, qd)
        fe.frontend()
        rec = fe.download_records()
        fe.close()
        for key in ("M", "J", "dq", "bias", "ee_pose"):
            g[key] = rec[key].astype(dt)
        tgt = rec["ee_pose"].copy()
        tgt[:, :, :3] += rngp.normal(0.0, 0.2, tgt[:, :, :3].shape)
        g["tgt_pose"] = tgt.astype(dt)
    if a.stress:
        rng = np.random.default_rng(4242 + sd)
        k = g["J"].shape[1]
        nrows = rng.integers(1, a.stress_rows + 1, size=B)
        for j in range(a.stress_rows):
            rows = rng.integers(0, k, size=B)
            f = np.where(nrows > j, 10.0 ** rng.uniform(-3.5, -1.5, size=B), 1.0)
            g["J"][np.arange(B), rows, :] = (g["J"][np.arange(B), rows, :] * f[:, None]).astype(dt)
    res = {}
    # the reference for float32 records is the generic kernel in float64 on the SAME (rounded) numbers
    g64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype == np.float32 else v) for k, v in g.items()}
    for name, kern, kdt, data in (("row16", _lib.KERNEL_ROW16, dt, g), ("generic", 1, np.float64, g64)):
        osc = BatchedOSC(lay, B, dtype=kdt, kernel=kern)
        osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
        # upload + step (not the one-call tick): uploaded records are probed for the kinematic tree's zero pattern, and physical
        # ones then go through the tree-structured form of the row16 kernel -- the form the sweep is meant to exercise
        osc.upload(data["M"], data["J"], data["dq"], data["bias"], data["ee_pose"], data.get("wrench"))
        osc.set_targets(data["tgt_pose"], data.get("tgt_vel"))
        res[name] = osc.step(return_flags=True)
        res[name + "_kernel"] = osc.kernel_name + ("+tree" if osc.slot_structure(0) else "")
        osc.close()
    (u, fl), (ug, flg) = res["row16"], res["generic"]
    err = np.max(np.abs(u.astype(np.float64) - ug), axis=1) / np.maximum(np.max(np.abs(ug), axis=1), 1e-300)
    over = np.nonzero(~(err <= a.tol))[0]
    n_in = 0
    for b in over:
        Mx, Minv, Mxi, det = osc_oracle.task_inertia(g["J"][b].astype(np.float64), g["M"][b].astype(np.float64))
        s = np.linalg.svd(Mxi, compute_uv=False)
        near = np.any(np.abs(s / s[0] / 1e-5 - 1.0) < a.band) if abs(det) < 1e-4 else not (s[-1] > 1e-12 * s[0])
        if not near:
            n_in += 1
            print(f"  IN DOMAIN seed {sd} b={b} err={err[b]:.3e} flags {fl[b]:#x}/{flg[b]:#x} det={det:.3e} "
                  f"tail s/(1e-5 smax) {np.array2string(s[-3:] / s[0] / 1e-5, precision=5)}", flush=True)
    tot += B
    bad_in += n_in
    bad_out += len(over) - n_in
    print(f"seed {sd}: {res['row16_kernel']} vs {res['generic_kernel']}, max rel diff {np.nanmax(err):.2e}, {len(over)} of {B} over {a.tol:g}, {n_in} of them inside the parity domain; "
          f"eigen-path {int(((fl & 4) != 0).sum())}, truncated {int(((fl & 8) != 0).sum())}", flush=True)
print(f"TOTAL {tot} instances: {bad_in} over 1e-5 inside the parity domain, {bad_out} outside (a singular value within 1 % of the cut)")
