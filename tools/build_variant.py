#!/usr/bin/env python3
"""A/B builds: the library with extra compiler definitions, next to the product build (which it never touches).
    python tools/build_variant.py NAME -DIRLOSC_R16_WAVES=3 -DIRLOSC_R16_PF=4 [--only tu_row16_f64 ...]
-> tools/_exp/libirlosc_NAME.so (objects under tools/_exp/obj_NAME/; `--only` recompiles just those units with the
definitions and takes the product build's objects for the rest).  Select it with IRLOSC_LIB=tools/_exp/libirlosc_NAME.so.
tools/_exp/ is git-ignored; the libraries travel to the GPU box with the snapshot, so delete them when done."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g                                   # noqa: E402


def main():
    args = sys.argv[1:]
    name, defs, only = args[0], [], []
    it = iter(args[1:])
    for a in it:
        if a == "--only":
            only = list(it)
        else:
            defs.append(a)
    out = os.path.join(ROOT, "tools", "_exp")
    objdir = os.path.join(out, f"obj_{name}")
    os.makedirs(objdir, exist_ok=True)
    g.build()                                                  # the product objects must be current
    objs, jobs = [], []
    for tu in g.translation_units():
        base = os.path.basename(tu)[:-4]
        if only and base not in only:
            objs.append(os.path.join(g.OBJDIR, base + ".o"))
            continue
        obj = os.path.join(objdir, base + ".o")
        objs.append(obj)
        jobs.append([g.HIPCC] + g.HIP_FLAGS + defs + ["-c", tu, "-o", obj])
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(lambda c: subprocess.check_call(c, cwd=g.CSRC), jobs))
    lib = os.path.join(out, f"libirlosc_{name}.so")
    subprocess.check_call([g.HIPCC] + g.LINK_FLAGS + objs + ["-o", lib], cwd=g.CSRC)
    print(lib)


if __name__ == "__main__":
    main()
