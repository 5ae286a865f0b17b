#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 --kernel-trace result: per kernel name the mean duration and the mean idle gap
before it (time since the previous kernel on the device ended).  usage: rocprof_gaps.py <results.db> [last N]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
last = int(sys.argv[2]) if len(sys.argv) > 2 else 400
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_c = "name" if "name" in cols else "kernel_name"
rows = db.execute(f"select {name_c}, start, end from kernels order by start").fetchall()[-last:]
agg = defaultdict(lambda: [0, 0.0, 0.0])
prev_end = None
for nm, st, en in rows:
    a = agg[nm[:70]]
    a[0] += 1
    a[1] += (en - st) / 1e3
    if prev_end is not None:
        a[2] += max(0, st - prev_end) / 1e3
    prev_end = en
span = (rows[-1][2] - rows[0][1]) / 1e3
print(f"# last {len(rows)} kernels, span {span:.1f} us")
for nm, (n, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{nm:70s} n={n:4d} avg_dur_us={d / n:9.2f} avg_gap_before_us={g / n:7.2f}")
