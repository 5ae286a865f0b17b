#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing, split at s_memtime stamps (the IRLOSC_TS phase marks).
usage: isa_phases.py <file.s> <mangled kernel name substring>"""
import re
import sys
from collections import Counter

path, sub = sys.argv[1:3]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sub in l and l.rstrip().endswith(tuple([":"])) or (l.startswith("_Z") and sub in l and ": ;" in l))
phases = [Counter()]
for l in lines[start + 1:]:
    t = l.strip()
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    if op == "s_endpgm":
        break
    if op in ("s_memtime", "s_memrealtime"):
        phases.append(Counter())
        continue
    if op.startswith("v_"):
        cat = "valu_pk" if op.startswith("v_pk_") else ("valu_dpp" if "dpp" in t or "quad_perm" in t or "row_" in t else "valu")
    elif op.startswith("s_waitcnt"):
        cat = "waitcnt"
    elif op.startswith("s_nop"):
        cat = "nop"
    elif op.startswith("s_"):
        cat = "salu"
    elif op.startswith("ds_"):
        cat = "lds"
    elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        cat = "vmem" if "lds" not in t else "dma"
    else:
        cat = "other"
    phases[-1][cat] += 1
    if cat == "nop":
        m = re.search(r"s_nop\s+(\d+)", t)
        phases[-1]["nop_cycles"] += int(m.group(1)) + 1
cats = ["valu", "valu_pk", "valu_dpp", "salu", "waitcnt", "nop", "nop_cycles", "lds", "dma", "vmem", "other"]
print("phase " + " ".join(f"{c:>9s}" for c in cats) + "     total")
tot = Counter()
for i, p in enumerate(phases):
    tot.update(p)
    print(f"{i:5d} " + " ".join(f"{p[c]:9d}" for c in cats) + f" {sum(v for k, v in p.items() if k != 'nop_cycles'):9d}")
print("  all " + " ".join(f"{tot[c]:9d}" for c in cats) + f" {sum(v for k, v in tot.items() if k != 'nop_cycles'):9d}")
