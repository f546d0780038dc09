"""CPU checks of the build recipe (content-hash staleness, dependency discovery) and of bench.py's own launcher
(`python bench.py --gpus 2` without torchrun) -- no GPU, no kernel launch."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bld():
    import importlib
    b = importlib.import_module("mujoco_drone_amd.build")
    b.build_library()
    return b


def test_every_include_of_the_sources_is_a_tracked_dependency(bld):
    deps = set(bld.dependencies())
    assert any(d.endswith("qd_contact.h") for d in deps)          # the header round 1's hand-written list had lost
    assert any(d.endswith(os.path.join("include", "qd.h")) for d in deps)
    seen, todo = set(), list(bld.sources())
    assert todo, "no translation units found"
    while todo:                                                   # transitive closure of the quoted includes
        f = todo.pop()
        if f in seen:
            continue
        seen.add(f)
        assert f in deps, "%s is compiled into libqd.so but not tracked by build.dependencies()" % os.path.relpath(f, ROOT)
        todo.extend(bld.included_files(f))
    assert len(seen) >= 12


def test_library_carries_the_hash_of_the_sources_it_was_built_from(bld):
    assert not bld.needs_build()
    want = bld.source_hash(bld._extra_flags())
    assert re.fullmatch(r"[0-9a-f]{64}", want)
    assert bld.embedded_hash() == want
    from mujoco_drone_amd import _lib
    assert _lib.lib().qd_source_hash().decode() == want


def test_stale_library_is_detected(bld, tmp_path):
    # same bytes with another hash inside = a library built from other sources
    blob = open(bld.LIB, "rb").read()
    k = blob.find(bld.HASH_TAG) + len(bld.HASH_TAG)
    fake = tmp_path / "libqd_stale.so"
    fake.write_bytes(blob[:k] + b"0" * 64 + blob[k + 64:])
    assert bld.embedded_hash(str(fake)) == "0" * 64 != bld.source_hash(bld._extra_flags())
    assert bld.embedded_hash(str(tmp_path / "missing.so")) is None
    # and the hash moves with any dependency's content
    h0 = bld.source_hash()
    assert bld.source_hash(("-DX",)) != h0


def _run_bench(args, timeout=240):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_launches_its_own_ranks():
    """SURVEY 8e / train_PPO.py:90-94 (8 samplers feed one learner): `bench.py --gpus 2` started plainly must run TWO ranks
    and report n_gpus = 2 with the trajectory all-gather populated (dry rehearsal: gloo, CPU tensors, no env stepping)"""
    p = _run_bench(["--gpus", "2", "--dry", "--envs", "64", "--fragment", "16", "--steps", "20", "--warmup", "5"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5 and out["dry_run"] is True
    assert out["scaling"] == "weak" and out["config"]["global_envs"] == 128
    tg = out["config"]["trajectory_all_gather"]
    assert tg is not None and tg["backend"] == "gloo" and tg["gathered_content_ok"] is True
    assert tg["all_gather_bytes_per_rank_per_fragment"] == 16 * 64 * ((22 + 4 + 1) * 4 + 1)
    assert tg["overlapped_gathers"] >= 1 and tg["all_gather_ms_per_fragment"] > 0


def test_bench_launcher_reports_a_failed_rank():
    # without --dry every rank needs a GPU: here they fail, and the launcher must say so with a non-zero exit
    p = _run_bench(["--gpus", "2", "--envs", "64", "--fragment", "16", "--steps", "2", "--warmup", "0"])
    assert p.returncode != 0
    assert "rank" in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_bench_single_rank_dry_line():
    p = _run_bench(["--dry", "--envs", "32", "--fragment", "8", "--steps", "20", "--warmup", "5"])
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["config"]["trajectory_all_gather"] is None and out["dry_run"] is True


def test_step_kernel_codegen_facts(bld):
    """Properties of the step kernels' machine code that are worth 0.2-0.3 us each of the 4 us step ("The start
    of the kernel", "Where the waits go") and that a compiler or source change can silently undo: the first loads leave on
    preloaded SGPRs, no FLAT access (a FLAT atomic counts on the LDS counter, and its mere presence changes where the compiler
    waits), three barriers, and NO wait for a global load once the first barrier is passed (it would wait for the state stores)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_isa_check", os.path.join(ROOT, "tools", "kernel_isa_check.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    coop = chk.kernel_facts("k_step_coop<1>")
    assert coop["kernarg_preload_dwords"] == 7, coop          # g (2), actions (2), npad, n, main_blocks
    assert coop["flat_memory_instructions"] == 0, coop
    assert coop["barriers"] == 3, coop
    assert coop["vmcnt_waits_after_first_barrier"] == 0, coop
    assert coop["global_atomics"] == 3, coop                  # pool_request, pool_count (main path), the sampler's counter update
    wide = chk.kernel_facts("k_step_wide<true, 256, 1>")
    assert wide["kernarg_preload_dwords"] == 7 and wide["flat_memory_instructions"] == 0, wide
    for pat in ("k_step<true, 64, 1>", "k_step<true, 64, 2>", "k_step<false, 64, 3>"):
        k = chk.kernel_facts(pat)
        assert k["kernarg_preload_dwords"] == 0 and k["flat_memory_instructions"] == 0, k   # struct-first signature, global accesses
