#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, TCC slot limits) into
profiles/hbm_traffic.json.  Units and gfx950 correction per /opt/skills/guides/MI355X_MICROARCH.md §HBM:
counters are in KiB; FETCH_SIZE reads exactly half of a wide coalesced stream on gfx950 -> doubled.
usage: pmc_traffic.py <fetch.db> <write.db> <kernel substring> <json key> <out.json> [<kernel-trace.db> <instances> <steps per launch>]
With the kernel trace of the same command the entry also gets the trace's average duration of the full-size launches
(rocprof_avg_us), which bench.py reports as roofline.frac_rocprof next to the live HIP-event figure."""
import json
import sqlite3
import sys


def avg(db, counter, sub):
    con = sqlite3.connect(db)
    rows = con.execute("select value from counters_collection where counter_name=? and kernel_name like ?",
                       (counter, f"%{sub}%")).fetchall()
    vals = [r[0] for r in rows]
    top = max(vals)                                   # the same kernel also runs shorter launches (a train of fewer steps,
    vals = [v for v in vals if v >= 0.9 * top]        # the single step of a download): keep the full-size launches only
    return sum(vals) / len(vals), len(vals)


def trace_avg(db, sub):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name_c = "name" if "name" in cols else "kernel_name"
    d = [(e - s) / 1e3 for s, e in con.execute(f"select start, end from kernels where {name_c} like ? order by start", (f"%{sub}%",))]
    med = sorted(d)[len(d) // 2]
    full = [v for v in d if v >= 0.5 * med]           # full trains only (a bench run also issues a few shorter launches; relative
                                                      # to the median: one slow outlier must not define what "full" means)
    full = full[-128:]                                # in time order, the steady state: the first launches after the idle set-up run at
                                                      # unsettled clocks (round 3's profiles of a 168-step command averaged mostly those)
    return sum(full) / len(full), len(full)


fetch_db, write_db, sub, key, out = sys.argv[1:6]
f, nf = avg(fetch_db, "FETCH_SIZE", sub)
w, nw = avg(write_db, "WRITE_SIZE", sub)
try:
    data = json.load(open(out))
except (OSError, ValueError):
    data = {}
data[key] = dict(kernel_filter=sub, fetch_size_kib_raw=f, write_size_kib=w, dispatches=[nf, nw],
                 correction="FETCH_SIZE x2 (gfx950 wide coalesced reads), WRITE_SIZE as reported",
                 traffic_bytes_per_launch=(2.0 * f + w) * 1024.0)
if len(sys.argv) >= 9:
    us, n = trace_avg(sys.argv[6], sub)
    data[key].update(rocprof_avg_us=us, rocprof_launches=n, instances=int(sys.argv[7]), steps_per_launch=int(sys.argv[8]))
json.dump(data, open(out, "w"), indent=1)
print(json.dumps(data[key]))
