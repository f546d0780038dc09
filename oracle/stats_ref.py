"""CPU float64 restatement of what the reference logs about a train batch.  TEST INFRASTRUCTURE ONLY, like the rest of
oracle/: the product never imports it.

column_stats follows MyCallbacks.on_learn_on_batch (custom_logging.py:9-31: np.min / np.max / np.mean / np.var over axis 0
of train_batch['obs'] and ['actions']); pinned by tests/golden/stats_vectors.npz, which tests/golden/make_stats_golden.py
produced by running that callback itself.  episode_stats follows the quantities training.py:16-22 reads from RLlib's
result dict (episode_reward_mean, episode_len_mean, sum(hist_stats.episode_reward) / sum(hist_stats.episode_lengths));
RLlib builds them from the per-step rewards and the truncation flags of vector_step (BaseDroneEnv.py:276-284)."""
import numpy as np


def column_stats(x):
    """[rows, cols] -> dict of float64 arrays (the reference computes the same in float32)"""
    x = np.asarray(x, dtype=np.float64)
    return {"min": x.min(axis=0), "max": x.max(axis=0), "mean": x.mean(axis=0), "var": x.var(axis=0)}


def episode_stats(reward, truncated, carry=None):
    """reward, truncated [T, N]; carry [N, 2] (return, length of the running episodes) -> (episode returns, episode lengths, carry),
    plain Python loops: small cases only"""
    reward, truncated = np.asarray(reward, dtype=np.float64), np.asarray(truncated)
    T, N = reward.shape
    carry = np.zeros((N, 2)) if carry is None else np.array(carry, dtype=np.float64)
    rets, lens = [], []
    for n in range(N):
        ret, ln = carry[n]
        for t in range(T):
            ret += reward[t, n]
            ln += 1
            if truncated[t, n]:
                rets.append(ret); lens.append(ln)
                ret, ln = 0.0, 0.0
        carry[n] = ret, ln
    return np.array(rets), np.array(lens), carry
