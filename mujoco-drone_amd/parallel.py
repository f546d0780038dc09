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
    """Per-rank rollout fragment [T, N, ...] the step kernel writes into directly (no staging copy)."""

    def __init__(self, T, n, obs_dim, device):
        self.T, self.n, self.D = T, n, obs_dim
        self.obs = torch.empty((T, n, obs_dim), dtype=torch.float32, device=device)
        self.actions = torch.empty((T, n, 4), dtype=torch.float32, device=device)
        self.rewards = torch.empty((T, n), dtype=torch.float32, device=device)
        self.truncated = torch.empty((T, n), dtype=torch.uint8, device=device)

    def tensors(self):
        return {"obs": self.obs, "actions": self.actions, "rewards": self.rewards, "truncated": self.truncated}

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.tensors().values())


class FragmentGather:
    """All-gather of one fragment per call.  Output layout: [world, T, N, ...] (rank-major), i.e. the
    learner sees world*N envs.  Output buffers are allocated once and reused."""

    def __init__(self, frag: FragmentBuffers, world):
        self.world = world
        self.out = {k: torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
                    for k, t in frag.tensors().items()}

    def __call__(self, frag: FragmentBuffers, async_op=False):
        if self.world == 1 or not dist.is_initialized():
            for k, t in frag.tensors().items():
                self.out[k][0].copy_(t)
            return self.out, []
        works = []
        for k, t in frag.tensors().items():
            # concatenated form [world*T, N, ...] of the same buffer: accepted by both RCCL and gloo
            flat = self.out[k].view((self.world * t.shape[0],) + tuple(t.shape[1:]))
            works.append(dist.all_gather_into_tensor(flat, t, async_op=async_op))
        return self.out, [w for w in works if w is not None]

    def learner_view(self):
        """[T, world*N, ...] views for the learner (env axis = rank-major concatenation)"""
        res = {}
        for k, t in self.out.items():
            w, T, n = t.shape[:3]
            res[k] = t.permute(1, 0, 2, *range(3, t.dim())).reshape(T, w * n, *t.shape[3:])
        return res
