#!/usr/bin/env python3
"""Single-call latency of the B = 1 drop-in path: BatchedOSC.tick (irlosc_tick: pack, one H2D copy, the step, one D2H
copy, one synchronisation) per kernel, and the three-call form it replaces (upload + set_targets + step)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from irl_control_amd import BatchedOSC, synth, _lib  # noqa: E402


def probe(B, dtype, kernel, label):
    lay, gains, g = synth.make_batch("k13", max(B, 16), seed=1, dtype=dtype)
    g = {k: (v[:B] if isinstance(v, np.ndarray) else v) for k, v in g.items()}
    osc = BatchedOSC(lay, B, dtype=dtype, kernel=kernel)
    osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
    a = (g["M"], g["J"], g["dq"], g["bias"], g["ee_pose"], g["tgt_pose"])
    for _ in range(20):
        osc.tick(*a)
    t = []
    for _ in range(300):
        t0 = time.perf_counter(); osc.tick(*a); t.append(time.perf_counter() - t0)
    t3 = []
    for _ in range(100):
        t0 = time.perf_counter()
        osc.upload(g["M"], g["J"], g["dq"], g["bias"], g["ee_pose"]); osc.set_targets(g["tgt_pose"]); osc.step()
        t3.append(time.perf_counter() - t0)
    print(f"B={B:5d} {label:28s} {osc.kernel_name:28s} tick median {np.median(t) * 1e6:6.1f} us  p99 {np.quantile(t, 0.99) * 1e6:6.1f} us"
          f"   | upload+targets+step median {np.median(t3) * 1e6:6.1f} us")
    osc.close()
    return float(np.median(t))


if __name__ == "__main__":
    for B in (1, 16, 256, 4096):
        probe(B, np.float64, _lib.KERNEL_AUTO, "f64 auto (row16)")
        probe(B, np.float64, _lib.KERNEL_GENERIC, "f64 generic")
        probe(B, np.float32, _lib.KERNEL_AUTO, "f32 records auto (row16 mixed)")
