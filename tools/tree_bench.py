#!/usr/bin/env python3
"""Dense records of PHYSICAL robot states (front-end output of random joint states, targets near the end effectors) through
the fp64 row16 kernel: the tree-structured form of the factorisation against the dense recursion, same records, A/B in
one process (IRLOSC_TREE is read when a context is created).
    python tools/tree_bench.py [--batch 65536] [--steps 256] [--reps 3] [--dtype f64|mixed] [--layout k13]
Prints the time per step of both forms, their largest difference, and both against the float64 oracle on a sample."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irl_control_amd import BatchedOSC, _lib, synth                 # noqa: E402
from irl_control_amd.rigid_body import RigidBodyModel              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--steps", type=int, default=256)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--slots", type=int, default=4)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--layout", default="k13")
ap.add_argument("--oracle", type=int, default=512, help="instances of slot 0 checked against the oracle (0: none)")
a = ap.parse_args()
dt = np.float64 if a.dtype == "f64" else np.float32
lay = synth.make_layout(a.layout)
model = RigidBodyModel.load("dual_ur5")
B = a.batch
_, gains, arr = synth.make_batch(a.layout, B, seed=7, dtype=dt)


def run(tree: bool):
    os.environ["IRLOSC_TREE"] = "1" if tree else "0"
    rng = np.random.default_rng(5)
    osc = BatchedOSC(lay, B, dtype=dt, n_slots=a.slots, kernel=_lib.KERNEL_ROW16)
    osc.set_model(model)
    osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
    tgt0 = None
    for s in range(a.slots):
        q, qd = model.random_state(rng, B)
        osc.upload_q(q, qd, slot=s)
        osc.frontend(slot=s)
        ee = osc.download_records(s, keys=("ee_pose",))["ee_pose"].astype(np.float64)
        tgt = synth.targets_near(ee, rng)
        osc.set_targets(tgt.astype(dt), None, slot=s)
        if s == 0:
            tgt0 = tgt
    assert all(osc.slot_structure(s) == tree for s in range(a.slots)), [osc.slot_structure(s) for s in range(a.slots)]
    osc.step_resident(16)
    best = 1e9
    for _ in range(a.reps):
        _, ms_step = osc.step_resident(a.steps)
        best = min(best, ms_step)
    u0, f0 = osc.step(slot=0, return_flags=True)
    rec = osc.download_records(0) if a.oracle else None
    osc.close()
    return best, u0, f0, rec, tgt0


ms_t, u_t, f_t, rec, tgt0 = run(True)
ms_d, u_d, f_d, _, _ = run(False)
print(f"B={B} {a.dtype} {a.layout}: tree {ms_t * 1e3:7.1f} us per step ({B / ms_t / 1e3:6.1f} M steps/s), "
      f"dense {ms_d * 1e3:7.1f} us per step ({B / ms_d / 1e3:6.1f} M steps/s)")
den = np.maximum(np.abs(u_d).max(axis=1, keepdims=True), 1e-300)
print("tree vs dense: max rel diff %.3g, flags equal: %s; eigen %.3f truncated %.3f" %
      (float((np.abs(u_t - u_d) / den).max()), bool(np.array_equal(f_t, f_d)), ((f_t & 4) != 0).mean(), ((f_t & 8) != 0).mean()))
if a.oracle:
    from oracle import osc_oracle                                   # the checker only
    n = min(a.oracle, B)
    r = {k: np.asarray(v[:n], dtype=np.float64) for k, v in rec.items()}
    ref = osc_oracle.generate_batch(lay.as_oracle_dict(), gains, r["M"], r["J"], r["dq"], r["bias"], r["ee_pose"],
                                    tgt0[:n].astype(dt).astype(np.float64))
    for nm, u in (("tree", u_t), ("dense", u_d)):
        err = np.abs(u[:n] - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1e-300)
        print(f"{nm} vs oracle on {n}: median {np.median(err):.3g} p99 {np.quantile(err, 0.99):.3g} max {err.max():.3g}")
