#!/usr/bin/env python3
"""Average every counter of a rocprofv3 --pmc pass per kernel.  usage: pmc_dump.py <pmc_results.db> [kernel substring]"""
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
con = sqlite3.connect(db)
acc = defaultdict(lambda: [0.0, 0])
for name, kern, val in con.execute("select counter_name, kernel_name, value from counters_collection"):
    if sub in kern:
        a = acc[(kern[:60], name)]
        a[0] += val
        a[1] += 1
for (kern, name), (s, n) in sorted(acc.items()):
    print(f"{kern:60s} {name:28s} n={n:4d} avg={s / n:.6g}")
