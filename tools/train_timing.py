#!/usr/bin/env python3
"""Untraced timing artefact of the headline kernel: N consecutive trains of the default bench workload
(65 536 physical records, k13, float64), each with its own HIP event pair and the wall clock stamped INSIDE the kernel
(irlosc_time_trains: first wave's start, last wave's end, s_memrealtime at 100 MHz), next to the wall-clock figure bench.py's
`ms_per_step` is made of (irlosc_step_resident between device synchronisations) -- one box, one process, one file.

    python tools/train_timing.py [--trains 256] [--batch 65536] [--out profiles/rXX_train_timing.json]

What the file lets a reader derive without any tracer:
  kernel_span_us   = isolated duration of a train's kernel  -> what a rocprofv3 kernel trace reports per dispatch, minus the
                     tracer's own per-dispatch cost (compare profiles/*_kernel_stats.txt of the same box)
  period_us        = start-to-start of consecutive trains   -> ms_per_step x steps_per_train of the bench line
  roofline fraction = algorithmic bytes per train / period (or span) / 8 TB/s."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                        # noqa: E402
from irl_control_amd import BatchedOSC, synth                       # noqa: E402
from irl_control_amd.rigid_body import RigidBodyModel               # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--trains", type=int, default=256)
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--slots", type=int, default=4)
ap.add_argument("--layout", default="k13")
ap.add_argument("--from-q", action="store_true", help="the fused path from joint coordinates instead of dense records")
ap.add_argument("--out", default=None)
a = ap.parse_args()

B = a.batch
lay = synth.make_layout(a.layout)
model = RigidBodyModel.load("dual_ur5")
osc = BatchedOSC(lay, B, dtype=np.float64, n_slots=a.slots)
osc.set_model(model)
_, gains, _ = synth.make_batch(a.layout, 2, seed=1)
osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
for s in range(a.slots):
    rng = np.random.default_rng(20241008 + 2000 + 17 * s + 500000)       # bench.py's physical workload, rank 0
    qpos, qvel = model.random_state(rng, B)
    osc.upload_q(qpos, qvel, slot=s)
    osc.frontend(slot=s)
    ee = osc.download_records(s, keys=("ee_pose",))["ee_pose"]
    osc.set_targets(synth.targets_near(ee, rng), None, slot=s)
spl = osc.steps_per_launch
run = osc.step_resident_from_q if a.from_q else osc.step_resident
run(1000)                                                            # clocks settle
steps = a.trains * spl
osc.device_sync()
t0 = time.perf_counter()
run(steps)
osc.device_sync()
wall = time.perf_counter() - t0
tt = osc.time_trains(a.trains, from_q=a.from_q)
if a.from_q:
    for s in range(a.slots):
        osc.frontend(slot=s)
osc.device_sync()
t0 = time.perf_counter()
run(steps)
osc.device_sync()
wall2 = time.perf_counter() - t0
bytes_train = bench.algorithmic_bytes(lay.n, lay.k, lay.ndev, lay.admittance, 8) * B * spl
summ = bench.train_summary(tt, spl)
out = {"what": __doc__.split("\n\n")[0], "kernel": (osc.from_q_name if a.from_q else osc.kernel_name + ("+tree" if osc.slot_structure(0) else "")),
       "instances": B, "steps_per_train": spl, "trains": a.trains,
       "wall_clock": {"ms_per_step_before": wall / steps * 1e3, "ms_per_step_after": wall2 / steps * 1e3,
                      "us_per_train_before": wall / a.trains * 1e6, "us_per_train_after": wall2 / a.trains * 1e6,
                      "note": "irlosc_step_resident of trains x steps_per_train steps between hipDeviceSynchronize, like bench.py's timed region; "
                              "run before and after the stamped trains"},
       "summary": summ,
       "algorithmic_bytes_per_train": None if a.from_q else bytes_train,
       "roofline_frac": None if a.from_q else {
           "from_wall_clock": bytes_train / (wall / a.trains) / 8e12, "from_period": bytes_train / (summ["period_us"]["median"] * 1e-6) / 8e12,
           "from_kernel_span": bytes_train / (summ["kernel_span_us"]["median"] * 1e-6) / 8e12},
       "per_train": {"event_pair_us": [round(float(v) * 1e3, 2) for v in tt[:, 0]], "start_us": [round(float(v), 2) for v in tt[:, 1]],
                     "end_us": [round(float(v), 2) for v in tt[:, 2]], "sclk_mhz": [round(float(v), 1) for v in tt[:, 3]]}}
osc.close()
txt = json.dumps(out)
if a.out:
    with open(a.out, "w") as f:
        f.write(txt + "\n")
s = out["summary"]
print(f"{out['kernel']}: wall clock {out['wall_clock']['us_per_train_before']:.1f} / {out['wall_clock']['us_per_train_after']:.1f} us per train, "
      f"period {s['period_us']['median']:.1f}, kernel span {s['kernel_span_us']['median']:.1f}, event pair {s['event_pair_us']['median']:.1f} us")
