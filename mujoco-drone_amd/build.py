"""Build recipe of the HIP library (libqd.so) -- `hipcc --offload-arch=gfx950`, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
.so travels to the GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB = os.environ.get("QD_LIB") or os.path.join(PKG_DIR, "libqd.so")  # QD_LIB: diagnostic builds only
SOURCES = ["qd_kernels.hip"]
HEADERS = ["qd_math.h", "qd_model.h", "qd_dynamics.h", "qd_obsrew.h", "qd_rng.h", "qd_pid.h", "qd_stats.h", "qd_policy.h", "qd_policy_dist.h", "qd_policy_static.h", "qd_policy_host.inc", os.path.join("..", "..", "include", "qd.h")]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into libqd.so next to this file."""
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "-O3", "-std=c++17", "--offload-arch=" + ARCH, "-shared", "-fPIC", "-fno-gpu-rdc",
           "-Wno-unused-result", "-o", LIB] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose="-v" in sys.argv))
