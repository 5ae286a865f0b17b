#!/usr/bin/env python3
"""Wide check of the lane-per-robot front end against the generic wave-per-robot kernel (an independent implementation of the
same recursions; both are held to the rigid-body oracle on small batches by the tests): every record of 65 536 random robots per
seed, joint angles over many turns, joint velocities up to +-20 rad/s.
    python tools/frontend_sweep.py [--seeds 4] [--dtype f64|f32]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irl_control_amd import BatchedOSC, synth                      # noqa: E402
from irl_control_amd.rigid_body import RigidBodyModel              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=4)
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--dtype", default="f64")
a = ap.parse_args()
dt = np.float64 if a.dtype == "f64" else np.float32
B = a.batch
model = RigidBodyModel.load("dual_ur5")
worst = {}
for layout in ("k13", "k12_admit", "k7"):
    lay = synth.make_layout(layout)
    for sd in range(a.seeds):
        rng = np.random.default_rng(31337 + sd)
        q = rng.uniform(-np.pi, np.pi, (B, lay.n)) * rng.choice([1.0, 1.0, 7.0, 300.0], size=(B, 1))
        qd = rng.normal(0.0, 1.0, (B, lay.n)) * rng.choice([0.0, 1.0, 5.0, 20.0], size=(B, 1))
        rec = {}
        for fe in ("lane", "generic"):
            os.environ["IRLOSC_FRONTEND"] = fe
            osc = BatchedOSC(lay, B, dtype=dt)
            osc.set_model(model)
            assert ("_lane_" in osc.frontend_name) == (fe == "lane")
            osc.upload_q(q, qd)
            osc.frontend()
            rec[fe] = osc.download_records()
            osc.close()
        line = []
        for k in ("M", "J", "dq", "bias", "ee_pose"):
            x, y = rec["lane"][k].astype(np.float64), rec["generic"][k].astype(np.float64)
            scale = np.abs(y).reshape(B, -1).max(axis=1) + 1e-300
            e = (np.abs(x - y).reshape(B, -1).max(axis=1) / scale)
            worst[k] = max(worst.get(k, 0.0), float(np.nanmax(e)))
            bad = int((~np.isfinite(x)).sum())
            line.append(f"{k} {np.nanmax(e):.2e}" + (f" ({bad} non-finite!)" if bad else ""))
        print(f"{layout} seed {sd}: max relative difference per record: " + ", ".join(line), flush=True)
print("WORST", {k: f"{v:.2e}" for k, v in worst.items()})
