#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (kernel trace) into a small text table:
per kernel name -> calls, total ms, avg us, min us, max us, VGPRs, LDS.  Usage:
    python tools/rocprof_summary.py gpurun_out/prof_x/x_results.db > profiles/r01_x_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    rows = db.execute("select * from kernels").fetchall()
    ix = {c: i for i, c in enumerate(cols)}
    name_c = "name" if "name" in ix else "kernel_name"
    agg = {}
    for r in rows:
        nm = r[ix[name_c]]
        dur = (r[ix["end"]] - r[ix["start"]]) / 1e3   # ns -> us
        a = agg.setdefault(nm, dict(n=0, tot=0.0, mn=1e30, mx=0.0, row=r))
        a["n"] += 1; a["tot"] += dur; a["mn"] = min(a["mn"], dur); a["mx"] = max(a["mx"], dur)
    tot_all = sum(a["tot"] for a in agg.values()) or 1.0
    print(f"# rocprofv3 --kernel-trace summary of {path}")
    if "grid_x" in ix:      # the fused train kernel is launched in several shapes (full train, first train, flush):
        by = {}             # break the dominant kernel down by grid size so like is compared with like
        for r in rows:
            key = (r[ix[name_c]], (r[ix["grid_x"]], r[ix["grid_y"]]) if "grid_y" in ix else r[ix["grid_x"]])
            a = by.setdefault(key, [0, 0.0])
            a[0] += 1; a[1] += (r[ix["end"]] - r[ix["start"]]) / 1e3
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}  extra")
    for nm, a in sorted(agg.items(), key=lambda kv: -kv[1]["tot"]):
        r = a["row"]
        extra = " ".join(f"{c}={r[ix[c]]}" for c in ("grid_size", "workgroup_size", "lds_size", "scratch_size",
                                                     "vgpr_count", "accum_vgpr_count", "sgpr_count",
                                                     "grid_x", "workgroup_x", "lds_block_size", "arch_vgpr_count")
                         if c in ix)
        print(f"{nm[:70]:70s} {a['n']:6d} {a['tot'] / 1e3:10.3f} {a['tot'] / a['n']:10.2f} {a['mn']:10.2f} "
              f"{a['mx']:10.2f} {100 * a['tot'] / tot_all:6.2f}  {extra}")
    if "grid_x" in ix:
        dom = max(agg.items(), key=lambda kv: kv[1]["tot"])[0]
        print(f"# dominant kernel by launch shape (grid_x = 64 x blocks, grid_y = steps of the train where it has one)")
        for (nm, gx), (n, tot) in sorted(by.items(), key=lambda kv: -kv[1][1]):
            if nm == dom:
                print(f"#   grid={str(gx):>16s} calls={n:5d} avg_us={tot / n:10.2f}")
        # in time order: the first launches after the idle set-up run at unsettled clocks (the power management needs ~50 ms of
        # load), so the steady state is what the LAST dispatches show -- and what a long untraced run measures
        seq = sorted(((r[ix["start"]], (r[ix["end"]] - r[ix["start"]]) / 1e3) for r in rows if r[ix[name_c]] == dom), key=lambda t: t[0])
        full = [d for _, d in seq if d >= 0.5 * sorted(d2 for _, d2 in seq)[len(seq) // 2]]
        for lab, part in (("first 32", full[:32]), ("last 128", full[-128:]), ("last 32", full[-32:])):
            if part:
                print(f"#   dominant kernel, full-size launches in time order, {lab:>8s}: n={len(part):4d} avg_us={sum(part) / len(part):10.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
