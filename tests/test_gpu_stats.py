"""GPU parity of the train-batch statistics (qd_column_stats / qd_episode_stats) against the numbers the reference's own
callback logs (tests/golden/stats_vectors.npz) and against the float64 oracle at full fragment size.
Tolerance: min / max are exact; the device sums in float64, so mean / var agree with the float64 oracle to 1e-12 relative and
with the reference's float32 numpy results to float32 rounding (2e-6 / 2e-5 relative)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def SG():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return np.load(os.path.join(HERE, "golden", "stats_vectors.npz"))


@pytest.mark.parametrize("tag", ["small", "batch"])
def test_callback_keys_and_values_vs_reference(SG, tag):
    from mujoco_drone_amd.custom_logging import BatchStatistics
    batch = {"obs": torch.tensor(SG[tag + "_obs"], device="cuda"), "actions": torch.tensor(SG[tag + "_actions"], device="cuda")}
    result = BatchStatistics().on_learn_on_batch(policy=None, train_batch=batch, result={})
    assert len(result) == 4 * (22 + 4)
    for what, cols in (("obs", 22), ("act", 4)):
        got = {s: np.array([result["%s_%s%d" % (s, what, i)] for i in range(cols)]) for s in ("min", "max", "mean", "var")}
        np.testing.assert_array_equal(got["min"], SG["%s_min_%s" % (tag, what)])
        np.testing.assert_array_equal(got["max"], SG["%s_max_%s" % (tag, what)])
        np.testing.assert_allclose(got["mean"], SG["%s_mean_%s" % (tag, what)], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(got["var"], SG["%s_var_%s" % (tag, what)], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("rows,cols", [(1, 1), (1, 64), (5, 3), (63, 22), (1000, 7), (4096 * 16 + 3, 23), (200001, 4), (50000, 37)])
def test_column_stats_vs_oracle_shapes(rows, cols):
    """ragged shapes: one row, one column, the widest supported matrix, row counts that do not fill the last wave"""
    from mujoco_drone_amd.custom_logging import BatchStatistics
    from oracle import stats_ref as S
    rng = np.random.default_rng(rows * 131 + cols)
    x = (rng.normal(size=(rows, cols)) * rng.uniform(0.1, 5, cols) + rng.uniform(-20, 20, cols)).astype(np.float32)
    got = BatchStatistics().column_stats(torch.tensor(x, device="cuda"))
    want = S.column_stats(x)
    np.testing.assert_array_equal(got["min"], want["min"])
    np.testing.assert_array_equal(got["max"], want["max"])
    np.testing.assert_allclose(got["mean"], want["mean"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(got["var"], want["var"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("offset", [1, 2, 3])
@pytest.mark.parametrize("rows,cols", [(1, 2), (777, 22), (4099, 5)])
def test_column_stats_of_unaligned_views(rows, cols, offset):
    """a matrix that does not start on a 16-byte boundary (a slice of a larger buffer) and whose size is not a multiple of four
    floats: the floats outside the 16-byte units are folded in by the final kernel"""
    from mujoco_drone_amd.custom_logging import BatchStatistics
    from oracle import stats_ref as S
    rng = np.random.default_rng(rows + cols + offset)
    flat = torch.tensor(rng.normal(size=rows * cols + 8).astype(np.float32) * 3 + 1, device="cuda")
    x = flat[offset:offset + rows * cols].view(rows, cols)
    assert x.data_ptr() % 16 == 4 * offset and x.is_contiguous()
    got = BatchStatistics().column_stats(x)
    want = S.column_stats(x.cpu().numpy())
    np.testing.assert_array_equal(got["min"], want["min"])
    np.testing.assert_array_equal(got["max"], want["max"])
    np.testing.assert_allclose(got["mean"], want["mean"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(got["var"], want["var"], rtol=1e-9, atol=1e-12)


def test_column_stats_full_fragment_deterministic_and_nan():
    """BASELINE fragment size: obs [1024, 4096, 22] (369 MB) in one pass; run-to-run bit-identical (fixed reduction order);
    a NaN anywhere makes that column's four numbers NaN like numpy's, and leaves the others alone"""
    from mujoco_drone_amd.custom_logging import BatchStatistics
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((1024, 4096, 22), generator=g, device="cuda") * 2.0 + 7.0
    st = BatchStatistics()
    a = st.column_stats_tensor(x)
    b = st.column_stats_tensor(x)
    assert torch.equal(a, b)
    xd = x.view(-1, 22).double()
    want = torch.stack([xd.min(0).values, xd.max(0).values, xd.mean(0), xd.var(0, unbiased=False)])
    assert torch.equal(a[:2], want[:2])
    assert torch.allclose(a[2:], want[2:], rtol=1e-9, atol=1e-12)
    x[517, 33, 5] = float("nan")
    c = st.column_stats_tensor(x)
    assert bool(torch.isnan(c[:, 5]).all()) and torch.equal(c[:, :5], a[:, :5]) and torch.equal(c[:, 6:], a[:, 6:])
    with pytest.raises(NotImplementedError):
        st.column_stats(torch.zeros((4, 65), device="cuda"))
    with pytest.raises(ValueError):
        st.column_stats(torch.zeros((0, 22), device="cuda"))


def test_episode_stats_vs_oracle_with_carry():
    from mujoco_drone_amd.custom_logging import EpisodeStatistics
    from oracle import stats_ref as S
    rng = np.random.default_rng(5)
    N, T = 300, 57
    es = EpisodeStatistics(N)
    carry = None
    for frag in range(3):
        reward = rng.normal(size=(T, N)).astype(np.float32)
        trunc = (rng.uniform(size=(T, N)) < 0.04).astype(np.uint8)
        if frag == 2:
            trunc[:] = 0                                            # a fragment in which no episode ends
        got = es.update(torch.tensor(reward, device="cuda"), torch.tensor(trunc, device="cuda"))
        rets, lens, carry = S.episode_stats(reward, trunc, carry)
        assert got["episodes"] == len(rets)
        np.testing.assert_allclose(es.carry.cpu().numpy(), carry, rtol=1e-12, atol=1e-12)
        if len(rets):
            np.testing.assert_allclose(got["episode_reward_mean"], rets.mean(), rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(got["episode_len_mean"], lens.mean(), rtol=1e-12)
            np.testing.assert_allclose(got["mean_action_reward"], rets.sum() / lens.sum(), rtol=1e-12, atol=1e-12)   # training.py:18
            np.testing.assert_allclose(got["episode_reward_std"], rets.std(), rtol=1e-9, atol=1e-12)
            assert got["episode_reward_min"] == rets.min() and got["episode_reward_max"] == rets.max()
            assert got["episode_len_min"] == lens.min() and got["episode_len_max"] == lens.max()
        else:
            assert np.isnan(got["episode_reward_mean"]) and np.isnan(got["episode_len_mean"]) and np.isnan(got["episode_reward_min"])


def test_statistics_of_a_real_rollout_fragment():
    """end to end on the env's own fragment: PID-flown drones, max_steps 64 -> every env ends 4 episodes of 64 steps in 256 steps;
    the column statistics of the fragment equal numpy's on the copied-back fragment"""
    from mujoco_drone_amd.custom_logging import BatchStatistics, EpisodeStatistics
    from mujoco_drone_amd.environments.BaseDroneEnv import base_config
    from mujoco_drone_amd.environments.observation_wrappers import LocalFrameRPYParamsEnv
    from mujoco_drone_amd.environments.rewards import distance_energy_reward
    n, T = 512, 256
    env = LocalFrameRPYParamsEnv(dict(base_config, num_drones=n, reward_fcn=distance_energy_reward, max_steps=64, auto_reset=True,
                                      random_start_pos=False, regen_env_at_steps=0))
    env.vector_reset_tensor()
    env.pid_reset()
    obs, rew, trunc = env.rollout_pid_tensor(T)
    info = EpisodeStatistics(n).update(rew, trunc)
    assert info["episodes"] == 4 * n and info["episode_len_mean"] == 64 and info["episode_len_min"] == 64 == info["episode_len_max"]
    r = rew.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(info["mean_action_reward"], r.mean(), rtol=1e-12)
    np.testing.assert_allclose(info["episode_reward_mean"], r.reshape(4, 64, n).sum(1).mean(), rtol=1e-12)
    got = BatchStatistics().column_stats(obs)
    x = obs.cpu().numpy().reshape(-1, obs.shape[-1]).astype(np.float64)
    np.testing.assert_array_equal(got["min"], x.min(0))
    np.testing.assert_allclose(got["mean"], x.mean(0), rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(got["var"], x.var(0), rtol=1e-8, atol=1e-12)
