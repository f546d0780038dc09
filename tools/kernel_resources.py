#!/usr/bin/env python3
"""List registers / LDS / scratch of every kernel in libqd.so (reads the gfx950 code object embedded in the library
with llvm-readelf; no GPU needed).      python tools/kernel_resources.py > profiles/r01_kernel_resources.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mujoco-drone_amd", "libqd.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
FILT = "c++filt"


def main():
    data = open(LIB, "rb").read()
    starts = [m.start() for m in re.finditer(b"\x7fELF", data)][1:]
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for k, at in enumerate(starts):
            path = os.path.join(tmp, "co%d.elf" % k)
            open(path, "wb").write(data[at:])
            out = subprocess.run([READELF, "--notes", path], capture_output=True, text=True).stdout
            cur = {}
            for line in out.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s+(.*)", line)
                if not m:
                    continue
                key, val = m.group(1), m.group(2).strip()
                if key == "agpr_count" and cur.get("name"):
                    rows.append(cur); cur = {}
                if key in ("name", "vgpr_count", "sgpr_count", "agpr_count", "group_segment_fixed_size", "private_segment_fixed_size",
                           "vgpr_spill_count", "sgpr_spill_count", "max_flat_workgroup_size"):
                    cur[key] = val
            if cur.get("name"):
                rows.append(cur)
    rows = [r for r in rows if "vgpr_count" in r]
    names = subprocess.run([FILT], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print("%-6s %-6s %-6s %-8s %-8s %-6s %s" % ("vgpr", "agpr", "sgpr", "lds(B)", "scratch", "spill", "kernel"))
    for r, n in sorted(zip(rows, names), key=lambda t: t[1]):
        n = re.sub(r"\(.*", "", n)
        print("%-6s %-6s %-6s %-8s %-8s %-6s %s" % (r.get("vgpr_count"), r.get("agpr_count", "0"), r.get("sgpr_count"), r.get("group_segment_fixed_size", "0"),
                                                    r.get("private_segment_fixed_size", "0"), r.get("vgpr_spill_count", "0"), n))


if __name__ == "__main__":
    sys.exit(main())
