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
