"""Build recipe of the HIP library (libqd.so) -- `hipcc --offload-arch=gfx950`, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
.so travels to the GPU box with the repository snapshot.

Staleness is decided by content, not by timestamps: the SHA-256 over every file the library is
compiled from (csrc/*.hip, *.h, *.inc and include/qd.h -- discovered, not listed by hand) is
compiled into the library (`qd_source_hash()`, also findable in the file as the text
"QD_SOURCE_HASH=<hex>"), and `needs_build()` compares it with the sources as they are now.
A prebuilt libqd.so that ships with a snapshot is therefore reused only if it was built from
exactly these sources.
"""
import glob
import hashlib
import os
import re
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.normpath(os.path.join(PKG_DIR, "..", "include"))
LIB = os.environ.get("QD_LIB") or os.path.join(PKG_DIR, "libqd.so")  # QD_LIB: diagnostic builds only
ARCH = "gfx950"
# the step kernels' leading scalar arguments arrive in SGPRs at wave launch instead of through a kernarg fetch at the top of the
# kernel (csrc/qd_kernels.hip, StepKernarg); part of the source hash like the sources themselves
CODEGEN_FLAGS = ("-mllvm", "-amdgpu-kernarg-preload-count=5")
# per translation unit, on top of CODEGEN_FLAGS (part of the source hash too).  qd_rollout_coop.hip: the SLP vectoriser packs the
# float32 3-vector arithmetic of the applied wrench into v_pk_*_f32 pairs and pays for every pair with a register move -- in the
# issue-bound waves of the persistent kernel that is a loss (same box, alternating runs: 1.485 -> 1.347 us per step at 4096 envs,
# config 5 at 8192 envs 1.946 -> 1.847); the per-step kernels of qd_kernels.hip measured the other way in round 2 and keep it.
UNIT_FLAGS = {"qd_rollout_coop.hip": ("-fno-slp-vectorize",), "qd_rollout_fused.hip": ("-fno-slp-vectorize",),
              "qd_rollout_fused32.hip": ("-fno-slp-vectorize",), "qd_rollout_lat.hip": ("-fno-slp-vectorize",)}
HASH_TAG = b"QD_SOURCE_HASH="


def sources():
    """translation units of the library"""
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def dependencies():
    """every file whose content reaches the compiler: the translation units, all headers / include fragments beside
    them, and the public header"""
    deps = set(sources())
    for pat in ("*.h", "*.inc"):
        deps.update(glob.glob(os.path.join(CSRC, pat)))
    deps.update(glob.glob(os.path.join(INCLUDE, "*.h")))
    return sorted(deps)


def included_files(path):
    """the quoted #include targets of one source file, resolved against its directory"""
    text = open(path, encoding="utf-8").read()
    return [os.path.normpath(os.path.join(os.path.dirname(path), m)) for m in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, re.M)]


def source_hash(extra_flags=()):
    h = hashlib.sha256()
    for d in dependencies():
        h.update(os.path.relpath(d, PKG_DIR).encode())
        h.update(b"\0")
        h.update(open(d, "rb").read())
        h.update(b"\0")
    h.update(" ".join(CODEGEN_FLAGS + tuple(extra_flags)).encode())
    for unit in sorted(UNIT_FLAGS):
        h.update(("%s:%s" % (unit, " ".join(UNIT_FLAGS[unit]))).encode())
    return h.hexdigest()


def embedded_hash(lib_path=None):
    """the source hash a built library carries (None if it has none or does not exist)"""
    lib_path = lib_path or LIB
    try:
        blob = open(lib_path, "rb").read()
    except OSError:
        return None
    k = blob.find(HASH_TAG)
    if k < 0:
        return None
    hx = blob[k + len(HASH_TAG):k + len(HASH_TAG) + 64]
    return hx.decode() if re.fullmatch(rb"[0-9a-f]{64}", hx) else None


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _extra_flags():
    return tuple(os.environ.get("QD_EXTRA_HIPCC_FLAGS", "").split())


def needs_build():
    return embedded_hash() != source_hash(_extra_flags())


def _closure(path, seen=None):
    """a translation unit and everything it includes (quoted includes, transitively)"""
    seen = set() if seen is None else seen
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for inc in included_files(path):
        _closure(inc, seen)
    return seen


def _object_key(cmd, src):
    """what one object depends on: its command line (without the output path) and the content of every file it includes"""
    h = hashlib.sha256(" ".join(cmd).encode())
    for d in sorted(_closure(src)):
        h.update(os.path.relpath(d, PKG_DIR).encode() + b"\0" + open(d, "rb").read() + b"\0")
    return h.hexdigest()[:32]


def _compile_and_link(out, hash_define, flags, verbose=False):
    """every translation unit to an object with its own flags (side by side), then one link.  Objects are kept in
    QD_OBJ_CACHE (a directory; unset: no cache) under a key of their command and inputs, so that an edit to one unit
    recompiles that unit only -- a developer convenience, the library's own staleness check is the source hash."""
    import tempfile
    hip = _hipcc()
    common = [hip, "-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result", *CODEGEN_FLAGS, *flags]
    # diagnostic variants only: QD_UNIT_FLAGS="qd_rollout_fused.hip:-DX=1,-DY qd_kernels.hip:-DZ" adds flags to single units
    unit_extra = {u: f.split(",") for u, f in (w.split(":", 1) for w in os.environ.get("QD_UNIT_FLAGS", "").split())}
    cache = os.environ.get("QD_OBJ_CACHE")
    if cache:
        os.makedirs(cache, exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="qd_build_") as tmpdir:
        procs, objs = [], []
        for src in sources():
            unit = os.path.basename(src)
            cmd = common + list(UNIT_FLAGS.get(unit, ())) + unit_extra.get(unit, [])
            if unit == "qd_source_hash.hip":
                cmd = cmd + [hash_define]
            obj = os.path.join(tmpdir, unit + ".o")
            kept = os.path.join(cache, "%s.%s.o" % (unit, _object_key(cmd, src))) if cache and not verbose else None
            if kept and os.path.exists(kept):
                objs.append(kept)
                continue
            cmd = cmd + ["-c", src, "-o", obj]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd), obj, kept))
            objs.append(obj)
        # every compiler is waited for before anything is raised: the temporary directory is deleted on the way out, and a sibling
        # still writing its object there would fail for a reason that is not its own
        failed = [(cmd, p.returncode) for cmd, p, _, _ in procs if p.wait() != 0]
        if failed:
            raise subprocess.CalledProcessError(failed[0][1], failed[0][0])
        for _, _, obj, kept in procs:
            if kept:
                shutil.copyfile(obj, kept + ".tmp")
                os.replace(kept + ".tmp", kept)
        subprocess.check_call([hip, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-fno-gpu-rdc", "-o", out] + objs)


def build_library(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into libqd.so next to this file (only if the sources changed, unless forced)."""
    if not force and not needs_build():
        return LIB
    if os.environ.get("QD_UNIT_FLAGS"):
        raise RuntimeError("QD_UNIT_FLAGS is for diagnostic variants (build_variant), not for the library")
    extra = _extra_flags()
    tmp = LIB + ".tmp.%d" % os.getpid()
    try:
        _compile_and_link(tmp, '-DQD_SOURCE_HASH="%s"' % source_hash(extra), extra, verbose)
        os.replace(tmp, LIB)   # atomic: a process that has the old file mapped keeps its copy
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


def build_variant(name, flags=()):
    """A diagnostic build beside the tests (tests/_build/libqd_<name>.so): the product's recipe plus `flags`, e.g.
    build_variant("diag", ["-DQD_STAMPS"]).  Loaded with QD_LIB=<path> (exempt from the source-hash check)."""
    out_dir = os.path.join(os.path.dirname(PKG_DIR), "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libqd_%s.so" % name)
    _compile_and_link(out, '-DQD_SOURCE_HASH="variant:%s"' % name, list(flags))
    return out


if __name__ == "__main__":
    # python -m mujoco_drone_amd.build [--force] [-v]   |   python mujoco-drone_amd/build.py --variant diag -DQD_STAMPS
    if "--variant" in sys.argv:
        k = sys.argv.index("--variant")
        print(build_variant(sys.argv[k + 1], sys.argv[k + 2:]))
    else:
        print(build_library(force="--force" in sys.argv, verbose="-v" in sys.argv))
