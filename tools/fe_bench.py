#!/usr/bin/env python3
"""Time the rigid-body front end alone (irlosc_frontend) at a few batch sizes: launches back to back, one sync.
    python tools/fe_bench.py [--batches 65536,16384,131072] [--reps 50] [--dtype f64]
IRLOSC_FRONTEND=generic selects the wave-per-instance kernel."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irl_control_amd import BatchedOSC, synth                      # noqa: E402
from irl_control_amd.rigid_body import RigidBodyModel              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="65536,16384,131072")
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--dtype", default="f64")
a = ap.parse_args()
dt = np.float64 if a.dtype == "f64" else np.float32
lay = synth.make_layout("k13")
model = RigidBodyModel.load("dual_ur5")
rng = np.random.default_rng(5)
for B in [int(x) for x in a.batches.split(",")]:
    osc = BatchedOSC(lay, B, dtype=dt, n_slots=2)
    osc.set_model(model)
    q, qd = model.random_state(rng, B)
    osc.upload_q(q, qd, slot=0)
    for _ in range(5):
        osc.frontend(slot=0)
    osc.device_sync()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        osc.frontend(slot=0)
    osc.device_sync()
    ms = (time.perf_counter() - t0) / a.reps * 1e3
    rec_bytes = (lay.n * lay.n + lay.k * lay.n + 2 * lay.n + 7 * lay.ndev) * np.dtype(dt).itemsize
    print(f"B={B:7d} {a.dtype}: {ms * 1e3:8.1f} us per launch, {B / ms / 1e3:7.1f} M instances/s, "
          f"{B * rec_bytes / ms / 1e6:7.1f} GB/s of records written", flush=True)
    osc.close()
