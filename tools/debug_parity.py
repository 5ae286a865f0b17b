#!/usr/bin/env python3
"""Find the instances of a synthetic batch where the GPU path and the oracle differ by more than 1e-5 and print why
(spectrum of J M^-1 J^T around the pinv cut, flags).  usage: debug_parity.py [mode] [B] [seed]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from irl_control_amd import BatchedOSC, synth, _lib
from oracle import osc_oracle

mode = sys.argv[1] if len(sys.argv) > 1 else "f64"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 20241008 + 2000
dt = np.float64 if mode == "f64" else np.float32
kern = 3 if mode == "mixed" else 0
lay, gains, g = synth.make_batch("k13", B, seed=seed, dtype=dt)
osc = BatchedOSC(lay, B, dtype=dt, kernel=kern)
osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
u, fl = osc.generate_batched(g["M"], g["J"], g["dq"], g["bias"], g["ee_pose"], g["tgt_pose"], return_flags=True)
print(osc.kernel_name)
g64 = {k: v.astype(np.float64) for k, v in g.items()}
cand = np.nonzero(fl & 4)[0]          # eigen-path instances
ref = osc_oracle.generate_batch(lay.as_oracle_dict(), gains, g64["M"], g64["J"], g64["dq"], g64["bias"], g64["ee_pose"],
                                g64["tgt_pose"], idx=cand)
err = np.max(np.abs(u[cand].astype(np.float64) - ref[cand]), axis=1) / np.max(np.abs(ref[cand]), axis=1)
print(f"{len(cand)} eigen-path instances, {int((err > 1e-5).sum())} over 1e-5")
for b, e in zip(cand[err > 1e-5][:40], err[err > 1e-5][:40]):
    Mx, Minv, Mxi, det = osc_oracle.task_inertia(g64["J"][b], g64["M"][b])
    s = np.linalg.svd(Mxi, compute_uv=False)
    r = s / s[0] / 1e-5
    print(f"b={b} err={e:.2e} flags={fl[b]:#x} det={det:.2e} smallest s/(1e-5 smax): {np.array2string(r[-4:], precision=4)}")
