"""Multi-GPU sharding of the env batch: one process per GPU, each rank owns a contiguous
block of envs with its own seed (seed + rank); the envs never exchange data.  The only
collective is the per-fragment all-gather that concatenates trajectories for the learner
(torch.distributed backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no collective code at all: its 8 RLlib rollout workers ship
SampleBatches through Ray's object store (train_PPO.py:90-94)."""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, world_size, local_rank).  Single-process runs return (0, 1, 0) without a group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("QD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_seed(seed, rank):
    """distinct Philox key per rank (seed 42 + rank)"""
    return int(seed) + int(rank)


def shard_bounds(total_envs, rank, world):
    """contiguous block [lo, hi) of a global env batch owned by `rank`"""
    per = (total_envs + world - 1) // world
    lo = min(rank * per, total_envs)
    return lo, min(lo + per, total_envs)


class FragmentBuffers:
    """Per-rank rollout fragment [T, N, ...] the step kernel writes into directly (no staging copy).  The four tensors are
    views of ONE contiguous slab (each section 256-byte aligned), so a fragment travels as a single collective."""

    ALIGN = 256

    def __init__(self, T, n, obs_dim, device, slab=None):
        self.T, self.n, self.D = T, n, obs_dim
        shapes = (("obs", (T, n, obs_dim), torch.float32), ("actions", (T, n, 4), torch.float32),
                  ("rewards", (T, n), torch.float32), ("truncated", (T, n), torch.uint8))
        self.sections, off = {}, 0
        for name, shape, dtype in shapes:
            nbytes = int(torch.tensor([], dtype=dtype).element_size())
            for x in shape:
                nbytes *= x
            self.sections[name] = (off, nbytes, shape, dtype)
            off = (off + nbytes + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.slab_bytes = off
        self.slab = torch.empty((off,), dtype=torch.uint8, device=device) if slab is None else slab
        assert self.slab.numel() == off and self.slab.dtype == torch.uint8
        for name, (o, nbytes, shape, dtype) in self.sections.items():
            setattr(self, name, self.slab[o:o + nbytes].view(dtype).view(shape))

    def tensors(self):
        return {"obs": self.obs, "actions": self.actions, "rewards": self.rewards, "truncated": self.truncated}

    def nbytes(self):
        """payload bytes (without the alignment padding between sections)"""
        return sum(t.numel() * t.element_size() for t in self.tensors().values())


class FragmentGather:
    """All-gather of one fragment per call: ONE collective over the fragment's slab (457 MB per rank at T=1024, N=4096,
    D=22 -- large messages are what xGMI's point-to-point links want).  Output layout: [world, T, N, ...] (rank-major),
    i.e. the learner sees world*N envs.  Output buffers are allocated once and reused."""

    def __init__(self, frag: FragmentBuffers, world):
        self.world = world
        self.out_slab = torch.empty((world, frag.slab_bytes), dtype=torch.uint8, device=frag.slab.device)
        self.parts = [FragmentBuffers(frag.T, frag.n, frag.D, frag.slab.device, slab=self.out_slab[r]) for r in range(world)]
        self.out = {}
        for name, (o, nbytes, shape, dtype) in frag.sections.items():
            # strided [world, ...] view over the gathered slabs (no copy): row r = rank r's section (slab sizes and section
            # offsets are multiples of 256 bytes, so the reinterpretation is exact)
            esz = self.parts[0].tensors()[name].element_size()
            flat = self.out_slab.view(dtype)                                        # [world, slab_bytes / esz]
            self.out[name] = flat[:, o // esz:(o + nbytes) // esz].view((world,) + tuple(shape))

    def __call__(self, frag: FragmentBuffers, async_op=False):
        if self.world == 1 or not dist.is_initialized():
            self.out_slab[0].copy_(frag.slab)
            return self.out, []
        work = dist.all_gather_into_tensor(self.out_slab.view(-1), frag.slab, async_op=async_op)
        return self.out, [work] if work is not None else []

    def learner_view(self):
        """[T, world*N, ...] views for the learner (env axis = rank-major concatenation)"""
        res = {}
        for k, t in self.out.items():
            w, T, n = t.shape[:3]
            res[k] = t.permute(1, 0, 2, *range(3, t.dim())).reshape(T, w * n, *t.shape[3:])
        return res
