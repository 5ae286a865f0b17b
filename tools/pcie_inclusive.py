#!/usr/bin/env python3
"""The rate when the boundary hands over HOST buffers every tick ("separately end-to-end with H2D/D2H"):
upload of the records + targets, one step, download of the torques, B = 65 536.  Never the bench's `value` (inputs resident)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irl_control_amd import BatchedOSC, synth          # noqa: E402

B = 65536
for dt in (np.float64, np.float32):
    lay, gains, arr = synth.make_batch("k13", B, seed=5, dtype=dt)
    osc = BatchedOSC(lay, B, dtype=dt)
    osc.set_gains(gains["kp"], gains["kv"], gains["ko"], gains["k"], gains["d"], gains["max_vel"], gains["null_kv"])
    args = (arr["M"], arr["J"], arr["dq"], arr["bias"], arr["ee_pose"], arr["tgt_pose"])
    for _ in range(3):
        osc.generate_batched(*args)
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        osc.generate_batched(*args)
    el = (time.perf_counter() - t0) / n
    nbytes = sum(a.nbytes for a in args) + B * lay.n * np.dtype(dt).itemsize
    print(f"{np.dtype(dt).name}: {el * 1e3:.1f} ms per tick of {B} robots from pageable host arrays = {B / el:.3g} steps/s, "
          f"{nbytes / el / 1e9:.1f} GB/s over PCIe ({osc.kernel_name})")
    osc.close()
