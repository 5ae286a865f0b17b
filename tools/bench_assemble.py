#!/usr/bin/env python3
"""Time irlosc_upload_raw at the bench batch size (run under rocprofv3 --kernel-trace to get the kernel alone)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irl_control_amd import BatchedOSC, _lib, synth

B, nv, ns = 65536, 37, 18
lay = synth.make_layout("k13")
rng = np.random.default_rng(0)
f = np.float32
arr = dict(qM=rng.normal(size=(B, nv, nv)).astype(f), qvel=rng.normal(size=(B, nv)).astype(f),
           qfrc_bias=rng.normal(size=(B, nv)).astype(f), jacp=rng.normal(size=(B, 3, 3, nv)).astype(f),
           jacr=rng.normal(size=(B, 3, 3, nv)).astype(f), ee_xpos=rng.normal(size=(B, 3, 3)).astype(f),
           ee_xquat=rng.normal(size=(B, 3, 4)).astype(f), site_xmat=rng.normal(size=(B, 3, 9)).astype(f),
           sensordata=rng.normal(size=(B, ns)).astype(f))
d = _lib.RawDesc()
d.nv, d.n_sensor = nv, ns
for p in range(32):
    d.joint_ids[p] = p if p < 25 else 0
    d.dq_src[p] = p if p < 25 else -1
for i in range(4):
    d.ft_force0[i], d.ft_torque0[i] = (6 * i, 6 * i + 3) if i < 2 else (-1, -1)
osc = BatchedOSC(lay, B, dtype=f)
for it in range(3):
    t0 = time.perf_counter()
    osc.upload_raw(d, **arr)
    dt = time.perf_counter() - t0
raw_bytes = sum(a.nbytes for a in arr.values())
print(f"upload_raw: {dt * 1e3:.2f} ms for {B} instances ({raw_bytes / 1e6:.0f} MB raw -> {B * 1067 * 4 / 1e6:.0f} MB records), "
      f"{raw_bytes / dt / 1e9:.1f} GB/s host-to-records")
osc.close()
