#!/usr/bin/env python3
"""Time the path from joint coordinates alone (irlosc_step_resident_from_q): trains back to back, HIP events.
    python tools/fromq_bench.py [--batch 65536] [--steps 64] [--reps 3] [--dtype f64|mixed] [--layout k13]
IRLOSC_FUSED=0 selects the two-kernel path through dense records (A/B)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irl_control_amd import BatchedOSC, _lib, synth                 # noqa: E402
from irl_control_amd.rigid_body import RigidBodyModel              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--slots", type=int, default=4)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--layout", default="k13")
a = ap.parse_args()
dt = np.float64 if a.dtype == "f64" else np.float32
lay = synth.make_layout(a.layout)
model = RigidBodyModel.load("dual_ur5")
rng = np.random.default_rng(5)
B = a.batch
osc = BatchedOSC(lay, B, dtype=dt, n_slots=a.slots, kernel=_lib.KERNEL_ROW16)
osc.set_model(model)
_, gains, arr = synth.make_batch(a.layout, B, seed=7, dtype=dt)
osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
for s in range(a.slots):
    q, qd = model.random_state(rng, B)
    osc.upload_q(q, qd, slot=s)
    osc.set_targets(arr["tgt_pose"], arr.get("tgt_vel"), slot=s)
print(osc.from_q_name, flush=True)
osc.step_resident_from_q(16)
for _ in range(a.reps):
    ms_total, ms_step = osc.step_resident_from_q(a.steps)
    print(f"B={B} {a.dtype} {a.layout}: {ms_step * 1e3:8.1f} us per step, {B / ms_step / 1e3:7.1f} M steps/s", flush=True)
u, fl = osc.download(B)
print("flags: eigen %.3f truncated %.3f nonfinite %d" % (((fl & 4) != 0).mean(), ((fl & 8) != 0).mean(), int(((fl & 64) != 0).sum())))
osc.close()
