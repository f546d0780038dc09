"""N>1 path on the CPU: two processes, gloo backend -- rank sharding, seeds and the per-fragment all-gather that
concatenates trajectories for the learner (RCCL all-gather on the GPU box)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from mujoco_drone_amd import parallel as par
    r, w, l = par.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    T, n, D = 5, 6, 22
    frag = par.FragmentBuffers(T, n, D, "cpu")
    g = torch.Generator().manual_seed(par.shard_seed(42, rank))
    frag.obs.copy_(torch.rand(frag.obs.shape, generator=g))
    frag.actions.copy_(torch.rand(frag.actions.shape, generator=g))
    frag.rewards.copy_(torch.rand(frag.rewards.shape, generator=g) + rank)
    frag.truncated.copy_((torch.rand(frag.truncated.shape, generator=g) > 0.5).to(torch.uint8))
    gather = par.FragmentGather(frag, world)
    out, works = gather(frag)
    for wk in works:
        wk.wait()
    ok = True
    for k, t in frag.tensors().items():
        ok &= bool(torch.equal(out[k][rank], t))              # my shard sits at index `rank`
    view = gather.learner_view()
    ok &= tuple(view["obs"].shape) == (T, world * n, D) and tuple(view["rewards"].shape) == (T, world * n)
    ok &= bool(torch.equal(view["obs"][:, rank * n:(rank + 1) * n], frag.obs))
    other = 1 - rank
    ok &= bool((view["rewards"][:, other * n:(other + 1) * n] >= other).all()) and bool(
        (view["rewards"][:, other * n:(other + 1) * n] < other + 1).all())
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


def test_fragment_all_gather_world_size_2():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_sharding_helpers():
    from mujoco_drone_amd import parallel as par
    assert [par.shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [par.shard_bounds(8192, r, 2) for r in range(2)] == [(0, 4096), (4096, 8192)]
    assert par.shard_seed(42, 3) == 45
    frag = par.FragmentBuffers(1024, 4096, 22, "meta")
    assert frag.nbytes() == 1024 * 4096 * ((22 + 4 + 1) * 4 + 1)     # 457 MB per rank per fragment
    single = par.FragmentGather(par.FragmentBuffers(2, 3, 4, "cpu"), 1)
    fb = par.FragmentBuffers(2, 3, 4, "cpu")
    fb.obs.fill_(1.5)
    out, _ = single(fb)
    assert float(out["obs"].min()) == 1.5
