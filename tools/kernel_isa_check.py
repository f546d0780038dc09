#!/usr/bin/env python3
"""Static facts about the step kernels' machine code, read from the gfx950 code object inside libqd.so (no GPU needed):
kernarg preload length (kernel descriptor), FLAT memory instructions, workgroup barriers, and `s_waitcnt vmcnt` instructions
after the first barrier.  DESIGN.md section 4 ("The start of the kernel", "Where the waits go") explains why each of them is worth
0.2-0.3 us of the 4 us step; tests/test_build_and_bench_cpu.py pins them.       python tools/kernel_isa_check.py"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mujoco-drone_amd", "libqd.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(lib=LIB, tmpdir=None):
    """path of the (first) device ELF embedded in the library, extracted into tmpdir"""
    data = open(lib, "rb").read()
    starts = [m.start() for m in re.finditer(b"\x7fELF", data)][1:]
    if not starts:
        raise RuntimeError("no embedded code object in %s" % lib)
    path = os.path.join(tmpdir, "qd_gfx950.elf")
    open(path, "wb").write(data[starts[0]:])
    return path


def symbols(elf):
    out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "--wide", elf], capture_output=True, text=True, check=True).stdout
    syms = {}
    for ln in out.splitlines():
        f = ln.split()
        if len(f) >= 8 and f[0].endswith(":") and f[0][:-1].isdigit():
            syms[f[7]] = (int(f[1], 16), int(f[2]), f[6])     # value, size, section index
    return syms


def sections(elf):
    out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-S", "--wide", elf], capture_output=True, text=True, check=True).stdout
    secs = {}
    for m in re.finditer(r"\[\s*(\d+)\]\s+(\S+)\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", out):
        secs[m.group(1)] = (m.group(2), int(m.group(3), 16), int(m.group(4), 16))   # name, address, file offset
    return secs


def kernel_facts(pattern, lib=LIB):
    """facts of the one kernel whose demangled name contains `pattern`"""
    with tempfile.TemporaryDirectory() as tmp:
        elf = code_object(lib, tmp)
        syms = symbols(elf)
        kds = [s for s in syms if s.endswith(".kd")]
        names = subprocess.run(["c++filt"], input="\n".join(k[:-3] for k in kds), capture_output=True, text=True).stdout.splitlines()
        hit = [(k, n) for k, n in zip(kds, names) if pattern in n]
        if len(hit) != 1:
            raise RuntimeError("%d kernels match %r: %s" % (len(hit), pattern, [n for _, n in hit][:4]))
        kd, name = hit[0]
        value, size, shndx = syms[kd]
        _, addr, off = sections(elf)[shndx]
        blob = open(elf, "rb").read()
        desc = blob[off + value - addr: off + value - addr + 64]
        preload = struct.unpack_from("<H", desc, 58)[0]            # amd_kernel_code kernarg_preload: length in bits 0-6
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + kd[:-3], elf], capture_output=True,
                             text=True, check=True).stdout
        ins = [ln.split("//")[0].strip() for ln in dis.splitlines() if re.match(r"^\s+[a-z_0-9]+", ln)]
        first_barrier = next((k for k, t in enumerate(ins) if t.startswith("s_barrier")), None)
        after = ins[first_barrier:] if first_barrier is not None else []
        return {"name": name, "kernarg_preload_dwords": preload & 0x7f, "instructions": len(ins),
                "flat_memory_instructions": sum(1 for t in ins if t.startswith("flat_")),
                "barriers": sum(1 for t in ins if t.startswith("s_barrier")),
                "vmcnt_waits_after_first_barrier": sum(1 for t in after if t.startswith("s_waitcnt") and "vmcnt" in t),
                "global_atomics": sum(1 for t in ins if t.startswith("global_atomic"))}


if __name__ == "__main__":
    for pat in sys.argv[1:] or ["k_step_coop<1>", "k_step_wide<true, 256, 1>", "k_step<true, 64, 1>", "k_step<true, 64, 2>"]:
        print(kernel_facts(pat))
