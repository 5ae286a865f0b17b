#!/usr/bin/env python3
"""Instruction mix of one kernel in a `hipcc -S --cuda-device-only` listing: fp64 arithmetic, AGPR spill moves, LDS, SALU, VMEM.
    python tools/isa_mix.py <file.s> <kernel-name substring>"""
import collections
import sys

path, sub = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = [i for i, ln in enumerate(lines) if ln.startswith("_Z") and ":" in ln and sub in ln.split(":")[0]][0]
end = [i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm")][0]
cnt = collections.Counter()
for ln in lines[start:end + 1]:
    ln = ln.strip()
    if not ln or ln[0] in ";." or ln.endswith(":"):
        continue
    cnt[ln.split()[0]] += 1
g = collections.Counter()
for op, c in cnt.items():
    if op.startswith("v_accvgpr"): g["v_accvgpr (AGPR spill moves)"] += c
    elif op.split("_e")[0] in ("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64"): g["fp64 arithmetic"] += c
    elif op.startswith("v_"): g["other VALU"] += c
    elif op.startswith("s_"): g["SALU / SMEM / waitcnt"] += c
    elif op.startswith("ds_"): g["LDS"] += c
    else: g["VMEM (" + op.split("_")[0] + ")"] += c
print(lines[start][:100])
print("total", sum(cnt.values()))
for k, v in g.most_common():
    print(f"  {k:32s} {v}")
print("  fp64 by opcode:", {k: v for k, v in cnt.items() if k.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64"))})
