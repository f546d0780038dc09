"""Train-batch statistics on the device: what the reference's custom_logging.py (MyCallbacks.on_learn_on_batch, :9-31) and
training.py (:16-22) compute on the host from a finished train batch, here over the rollout fragments where they lie in HBM.

    stats = BatchStatistics()
    stats.on_learn_on_batch(train_batch={'obs': obs[T,N,D], 'actions': actions[T,N,4]}, result=result)   # same keys as the reference
    episodes = EpisodeStatistics(num_envs)
    info = episodes.update(reward[T,N], truncated[T,N])     # 'episode_reward_mean', 'episode_len_mean', ... (RLlib's names)

The weight / gradient norms of MyCallbacks.on_train_result belong to the learner and are not part of this package."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class BatchStatistics:
    """per-column min / max / mean / population variance of device matrices (qd_column_stats): one streaming pass, float64
    sums in a fixed order"""

    def __init__(self):
        self.lib = L.lib()
        self._ws = {}

    def column_stats_tensor(self, x):
        """x: float32 CUDA tensor [..., cols] (contiguous) -> float64 CUDA tensor [4, cols] = min, max, mean, var; no sync"""
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
            raise ValueError("column statistics take a contiguous float32 CUDA tensor")
        cols = int(x.shape[-1])
        rows = x.numel() // max(cols, 1)
        key = (x.device, cols)
        if key not in self._ws:
            nbytes = self.lib.qd_column_stats_workspace_bytes(cols)   # 0 for an unsupported width: qd_column_stats reports it
            self._ws[key] = torch.empty(max(nbytes, 8) // 8, dtype=torch.float64, device=x.device)
        ws = self._ws[key]
        out = torch.empty((4, cols), dtype=torch.float64, device=x.device)
        L.check(self.lib.qd_column_stats(_ptr(x), rows, cols, _ptr(out), _ptr(ws), ws.numel() * 8, _stream(x.device)))
        return out

    def column_stats(self, x):
        """-> dict of float64 numpy arrays 'min', 'max', 'mean', 'var' (one host sync)"""
        out = self.column_stats_tensor(x).cpu().numpy()
        return {"min": out[0], "max": out[1], "mean": out[2], "var": out[3]}

    def on_learn_on_batch(self, *, policy=None, train_batch, result, **kwargs):
        """custom_logging.py:9-31 with the same result keys: 'min_obs%d', 'max_obs%d', 'mean_obs%d', 'var_obs%d' per observation
        column and 'min_act%d' ... per action column"""
        pending = [(name, self.column_stats_tensor(train_batch[key])) for key, name in (("obs", "obs"), ("actions", "act"))]
        for name, dev in pending:                      # both passes are enqueued before the first read-back
            s = dev.cpu().numpy()
            for i in range(s.shape[1]):
                result['min_%s%d' % (name, i)] = s[0, i]
                result['max_%s%d' % (name, i)] = s[1, i]
                result['mean_%s%d' % (name, i)] = s[2, i]
                result['var_%s%d' % (name, i)] = s[3, i]
        return result


def merge_column_stats(parts, rows):
    """Combine the column statistics of several shards of one train batch (e.g. one per rank after the fragment all-gather) into
    those of the whole batch: `parts` = dicts / [4, cols] arrays (min, max, mean, var) as `BatchStatistics.column_stats` returns,
    `rows` = the shards' row counts.  Host arithmetic on a few dozen numbers (parallel-variance formula), float64."""
    arr = [np.stack([p["min"], p["max"], p["mean"], p["var"]]) if isinstance(p, dict) else np.asarray(p, dtype=np.float64) for p in parts]
    n = np.asarray(rows, dtype=np.float64)
    if len(arr) != len(n) or len(arr) == 0 or np.any(n <= 0):
        raise ValueError("merge_column_stats needs one positive row count per shard")
    total = n.sum()
    mean = sum(a[2] * k for a, k in zip(arr, n)) / total
    var = sum((a[3] + (a[2] - mean) ** 2) * k for a, k in zip(arr, n)) / total
    return {"min": np.minimum.reduce([a[0] for a in arr]), "max": np.maximum.reduce([a[1] for a in arr]), "mean": mean, "var": var}


class EpisodeStatistics:
    """episode returns and lengths from rollout fragments (qd_episode_stats); the running episode of every env is carried
    from one fragment to the next"""

    FIELDS = ("episodes", "sum_reward", "sum_length", "sum_reward_sq", "episode_reward_min", "episode_reward_max",
              "episode_len_min", "episode_len_max", "episode_reward_mean", "episode_len_mean", "mean_action_reward",
              "episode_reward_std")

    def __init__(self, num_envs, device="cuda:0"):
        self.lib = L.lib()
        self.n, self.device = int(num_envs), torch.device(device)
        self.carry = torch.zeros((self.n, 2), dtype=torch.float64, device=self.device)
        self._ws = torch.empty(max(self.lib.qd_episode_stats_workspace_bytes(self.n), 8) // 8, dtype=torch.float64, device=self.device)

    def reset(self):
        self.carry.zero_()

    def update_tensor(self, reward, truncated):
        """reward float32 [T, N], truncated uint8 [T, N] on the device -> float64 CUDA tensor [12] (FIELDS); no sync"""
        if tuple(reward.shape) != tuple(truncated.shape) or reward.dim() != 2 or reward.shape[1] != self.n:
            raise ValueError("reward / truncated must be [T, %d]" % self.n)
        if reward.dtype != torch.float32 or truncated.dtype != torch.uint8 or not (reward.is_contiguous() and truncated.is_contiguous()):
            raise ValueError("reward must be contiguous float32, truncated contiguous uint8")
        out = torch.empty(12, dtype=torch.float64, device=self.device)
        L.check(self.lib.qd_episode_stats(_ptr(reward), _ptr(truncated), int(reward.shape[0]), self.n, _ptr(self.carry), _ptr(out),
                                          _ptr(self._ws), self._ws.numel() * 8, _stream(self.device)))
        return out

    def update(self, reward, truncated):
        """-> dict with RLlib's result names (episode_reward_mean / _min / _max, episode_len_mean, episodes) plus training.py:18's
        mean reward per action ('mean_action_reward'); one host sync"""
        v = self.update_tensor(reward, truncated).cpu().numpy()
        d = dict(zip(self.FIELDS, (float(x) for x in v)))
        d["episodes"] = int(d["episodes"])
        return d
