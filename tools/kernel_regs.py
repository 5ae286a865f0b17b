#!/usr/bin/env python3
"""Register / scratch / LDS figures of every kernel in libirlosc.so (code-object notes of the shipped library).

    python tools/kernel_regs.py [filter]      # e.g. row16, frontend_lane

The gfx950 code objects are cut out of the .so's offload bundles, llvm-readelf --notes prints the AMDGPU
metadata (.vgpr_count, .agpr_count, .vgpr_spill_count, .private_segment_fixed_size = scratch bytes per lane,
.group_segment_fixed_size = static LDS bytes per workgroup)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("IRLOSC_LIB", os.path.join(ROOT, "irl_control_amd", "libirlosc.so"))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(lib):
    """(offset, size) of every gfx950 code object: the .hip_fatbin section is a run of uncompressed clang offload bundles
    (magic, u64 entry count, then per entry u64 offset / u64 size / u64 triple length / triple)."""
    import struct
    with open(lib, "rb") as f:
        blob = f.read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    for m in re.finditer(magic, blob):
        base = m.start()
        pos = base + len(magic)
        (n,) = struct.unpack_from("<Q", blob, pos)
        pos += 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, pos)
            pos += 24
            triple = blob[pos:pos + tl].decode()
            pos += tl
            if "gfx950" in triple and size:
                yield base + off, size


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    rows = []
    with open(LIB, "rb") as f:
        blob = f.read()
    for off, size in code_objects(LIB):
        with tempfile.NamedTemporaryFile(suffix=".co") as t:
            t.write(blob[off:off + size])
            t.flush()
            notes = subprocess.run([READELF, "--notes", t.name], capture_output=True, text=True).stdout
        cur = {}
        for ln in notes.splitlines():
            ln = ln.strip()
            m = re.match(r"-?\s*\.(\w+):\s+(.*)", ln)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip("'\"")
            if k == "agpr_count" and cur.get("name"):      # a new kernel record starts with .agpr_count in this LLVM's ordering
                pass
            cur[k] = v
            if k == "wavefront_size":                        # last field of a kernel record
                if "name" in cur:
                    rows.append(cur)
                cur = {}
    for r in sorted(rows, key=lambda r: r.get("name", "")):
        nm = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        if flt and flt not in nm:
            continue
        nm = re.sub(r"irlosc::", "", nm)
        print(f"vgpr {r.get('vgpr_count', '?'):>3} agpr {r.get('agpr_count', '?'):>3} sgpr {r.get('sgpr_count', '?'):>3} "
              f"spill {r.get('vgpr_spill_count', '?'):>3} scratch {r.get('private_segment_fixed_size', '?'):>4} "
              f"lds {r.get('group_segment_fixed_size', '?'):>6}  {nm[:150]}")


if __name__ == "__main__":
    main()
