"""The env driven the way RLlib drives it ("train_PPO.py consumes it unchanged", north_star).

ray is not installed here, so the consumer is a 30-line stand-in for what RLlib's sampler does with a VectorEnv
(ray/rllib/env/vector_env.py VectorEnvWrapper [3P]: poll -> vector_reset once, then per step send_actions -> vector_step,
and try_reset -> reset_at(i) for every env whose episode ended), fed with an EnvContext-like config: a dict that carries
`worker_index` as an ATTRIBUTE, as the reference reads it (BaseDroneEnv.py:62 getattr, :113 config.get)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qd():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mujoco_drone_amd import _lib
    _lib.lib()  # fail loudly if the HIP library is missing
    return _lib


class FakeEnvContext(dict):
    """ray.rllib.env.env_context.EnvContext [3P]: the env_config dict plus worker bookkeeping attributes"""

    def __init__(self, env_config, worker_index, vector_index=0, remote=False, num_workers=8, recreated_worker=False):
        dict.__init__(self, env_config)
        self.worker_index, self.vector_index, self.remote = worker_index, vector_index, remote
        self.num_workers, self.recreated_worker = num_workers, recreated_worker


def sampler_loop(env, policy, steps):
    """what a rollout worker does with one VectorEnv: returns the per-step records it would put into SampleBatches"""
    n = env.num_envs
    obs, infos = env.vector_reset(seeds=[None] * n, options=[None] * n)
    assert isinstance(obs, list) and len(obs) == n and isinstance(infos, list) and len(infos) == n
    records = []
    for t in range(steps):
        actions = [policy(o) for o in obs]
        new_obs, rewards, terminateds, truncateds, infos = env.vector_step(actions)
        records.append((obs, actions, new_obs, rewards, terminateds, truncateds, infos))
        obs = list(new_obs)
        for i in range(n):
            if terminateds[i] or truncateds[i]:               # episode over: RLlib asks for this sub-env's reset observation
                ob_i, info_i = env.reset_at(i, seed=None, options=None)
                records[-1] = records[-1] + ((i, ob_i, info_i),)
                obs[i] = ob_i
    return records


def make(worker_index, as_key=False, **over):
    from mujoco_drone_amd.environments.BaseDroneEnv import base_config
    from mujoco_drone_amd.environments.observation_wrappers import LocalFrameRPYParamsEnv
    from mujoco_drone_amd.environments.rewards import distance_energy_reward
    cfg = dict(base_config)   # train_RMA.py:66-75 shape, shortened: 16 drones, episodes of <= 5 steps, regen every 8 steps
    cfg.update(num_drones=16, random_params=True, param_difficulty=1, state_difficulty=0.2, max_steps=5, regen_env_at_steps=8,
               reward_fcn=distance_energy_reward)
    cfg.update(over)
    if as_key:
        cfg["worker_index"] = worker_index
    return LocalFrameRPYParamsEnv(FakeEnvContext(cfg, worker_index))


def test_sampler_loop_types_shapes_and_reference_quirks(qd):
    env = make(worker_index=3)
    n, D, regen = env.num_envs, 22, env.regen_env_at_steps
    assert n == 16 and regen == 8
    assert env.observation_space.shape == (D,) and env.observation_space.dtype == np.float64
    assert env.action_space.shape == (4,) and float(env.action_space.low[0]) == 0.0 and float(env.action_space.high[0]) == 1.0
    rng = np.random.default_rng(0)
    p_first = env.drone_params[0]
    records = sampler_loop(env, lambda o: rng.uniform(0.3, 0.7, 4), 2 * regen)
    assert len(records) == 2 * regen
    for t, rec in enumerate(records, start=1):
        obs, actions, new_obs, rewards, terminateds, truncateds, infos = rec[:7]
        resets = rec[7:]
        assert isinstance(new_obs, list) and len(new_obs) == n
        assert all(isinstance(o, np.ndarray) and o.shape == (D,) and o.dtype == np.float64 and np.isfinite(o).all() for o in new_obs)
        assert isinstance(rewards, list) and len(rewards) == n and all(isinstance(r, float) for r in rewards)
        assert isinstance(terminateds, list) and terminateds == [False] * n          # out-of-bounds / max_steps end as truncated (C-5)
        assert isinstance(infos, list) and len(infos) == n and all(i == {} for i in infos)
        if t % regen == 0:
            # BaseDroneEnv.py:289-292: the regen step answers with np.ones(N, bool) -- an ndarray, not a list -- and every
            # sub-env is then reset_at() by the sampler on top of the full reset the env already did
            assert isinstance(truncateds, np.ndarray) and truncateds.dtype == bool and truncateds.shape == (n,) and truncateds.all()
            assert len(resets) == n
        else:
            assert isinstance(truncateds, list) and all(isinstance(x, bool) for x in truncateds)
            assert len(resets) == sum(truncateds)
        for (i, ob_i, info_i) in resets:
            assert isinstance(ob_i, np.ndarray) and ob_i.shape == (D,) and info_i == {}
            # QUIRK C-1: reset_at answers with the observation of BEFORE the reset (stale self.states)
            np.testing.assert_array_equal(ob_i, new_obs[i])
    # every episode is at most max_steps long: with max_steps = 5 and regen at 8 the 5th step truncates everybody
    assert records[4][5] == [True] * n
    assert env.drone_params[0] != p_first                    # regen drew new parameters (BaseDroneEnv.py:298-310)
    assert env.total_steps == 0 and int(env.num_steps.max()) == 0


def test_worker_index_attribute_does_not_reach_the_seed(qd):
    """QUIRK C-4: RLlib's EnvContext.worker_index is an attribute, the reference looks it up with config.get(): every rollout
    worker seeds with 42 and draws the same parameters and starts; only a dict KEY changes the seed"""
    a, b, c = make(worker_index=1), make(worker_index=5), make(worker_index=5, as_key=True)
    assert a.seed_value == b.seed_value == 42 and c.seed_value == 48
    oa, _ = a.vector_reset()
    ob, _ = b.vector_reset()
    oc, _ = c.vector_reset()
    assert a.drone_params == b.drone_params and a.drone_params != c.drone_params
    np.testing.assert_array_equal(np.array(oa), np.array(ob))
    assert not np.allclose(np.array(oa), np.array(oc))
    act = [np.full(4, 0.5)] * a.num_envs
    for _ in range(3):
        ra, rb = a.vector_step(act), b.vector_step(act)
        np.testing.assert_array_equal(np.array(ra[0]), np.array(rb[0]))
        assert ra[1] == rb[1]
