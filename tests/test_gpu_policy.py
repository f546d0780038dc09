"""GPU parity of the on-device policy inference (qd_policy_*) against the outputs of the reference's own
model classes (tests/golden/policy_vectors.npz) and against the float64 policy oracle on larger batches.
Tolerance: the reference computes in float32; the device's f32 MFMA chain and a fast tanh differ from it by a few
1e-6 on O(1) logits -> 2e-5 absolute."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
HERE = os.path.dirname(os.path.abspath(__file__))
TAGS = {"rma_full": "RMA_full", "rma_model": "RMA_model", "simple_mlp": "SimpleMLPmodel"}
ALL_TAGS = dict(TAGS, custom_mlp="CustomMLP")


@pytest.fixture(scope="module")
def PG():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return np.load(os.path.join(HERE, "golden", "policy_vectors.npz"))


def weights_of(PG, tag):
    return {k: PG[tag + "/" + k] for k in PG[tag + "_keys"]}


@pytest.fixture(params=["specialised", "interpreter"])
def kernel(request, monkeypatch):
    """both kernels behind qd_policy_*: the compile-time specialisation the library picks for the reference's networks
    and the generic layer-program interpreter (forced through QD_POLICY_GENERIC, read at qd_policy_create)"""
    if request.param == "interpreter":
        monkeypatch.setenv("QD_POLICY_GENERIC", "1")
    else:
        monkeypatch.delenv("QD_POLICY_GENERIC", raising=False)
    return request.param


@pytest.mark.parametrize("tag", list(ALL_TAGS))
def test_policy_forward_vs_reference_models(PG, tag, kernel):
    from mujoco_drone_amd.policy import DevicePolicy
    pol = DevicePolicy(ALL_TAGS[tag], weights_of(PG, tag))
    assert (pol.kernel > 0) == (kernel == "specialised")
    obs, prev = torch.tensor(PG["obs"], device="cuda"), torch.tensor(PG["prev_actions"], device="cuda")
    act, logits, value = pol.forward(obs, prev, want_logits=True, want_value=True)
    np.testing.assert_allclose(logits.cpu().numpy(), PG[tag + "_logits"], atol=2e-5)
    np.testing.assert_allclose(value.cpu().numpy(), PG[tag + "_value"], atol=2e-5)
    np.testing.assert_allclose(act.cpu().numpy(), PG[tag + "_action"], atol=1e-5)
    # actions only (the rollout configuration) gives the same actions
    np.testing.assert_array_equal(pol.forward(obs, prev).cpu().numpy(), act.cpu().numpy())


@pytest.mark.parametrize("n", [1, 17, 4096 + 5])
def test_policy_forward_vs_oracle_ragged_batches(PG, n, kernel):
    """batch sizes that are not multiples of the 16-env tile; previous actions absent / zeroed at episode starts"""
    from mujoco_drone_amd.policy import DevicePolicy
    from oracle import policy_ref as P
    rng = np.random.default_rng(n)
    w = weights_of(PG, "rma_full")
    pol = DevicePolicy("RMA_full", w)
    obs = rng.normal(scale=1.5, size=(n, 22)).astype(np.float32)
    obs[:, 16:] = (np.array([1, 0.17, 7, 0.01, 1.2, 0.3]) * (1 + 0.1 * rng.normal(size=(n, 6)))).astype(np.float32)
    prev = rng.uniform(0, 1, (n, 4)).astype(np.float32)
    tr = (rng.uniform(size=n) < 0.3).astype(np.uint8)
    act, logits, value = pol.forward(torch.tensor(obs, device="cuda"), torch.tensor(prev, device="cuda"),
                                     torch.tensor(tr, device="cuda"), want_logits=True, want_value=True)
    wl, wv = P.rma_full(w, obs, prev * (1 - tr[:, None]))
    np.testing.assert_allclose(logits.cpu().numpy(), wl, atol=2e-5)
    np.testing.assert_allclose(value.cpu().numpy(), wv, atol=2e-5)
    np.testing.assert_allclose(act.cpu().numpy(), P.beta_mean_action(wl), atol=1e-5)
    a0 = pol.forward(torch.tensor(obs, device="cuda"))              # no previous action at all = zeros
    wl0, _ = P.rma_full(w, obs, np.zeros((n, 4)))
    np.testing.assert_allclose(a0.cpu().numpy(), P.beta_mean_action(wl0), atol=1e-5)


def test_policy_closed_loop_rollout(PG, kernel):
    """qd_rollout_policy == T x (qd_policy_forward, qd_step): observations feed the policy, its actions feed the env and
    come back as prev_actions, zeroed where the env was just re-sampled; and against the float64 oracle policy fed
    with the device's observations"""
    from mujoco_drone_amd.policy import DevicePolicy
    from mujoco_drone_amd.environments.BaseDroneEnv import base_config
    from mujoco_drone_amd.environments.observation_wrappers import LocalFrameRPYParamsEnv
    from mujoco_drone_amd.environments.rewards import distance_energy_reward
    from oracle import policy_ref as P
    w = weights_of(PG, "rma_full")
    pol = DevicePolicy("RMA_full", w)
    cfg = dict(base_config, num_drones=200, reward_fcn=distance_energy_reward, random_params=True, param_difficulty=1,
               state_difficulty=0.2, max_steps=9, auto_reset=True)
    T = 30
    e1, e2 = LocalFrameRPYParamsEnv(cfg), LocalFrameRPYParamsEnv(cfg)
    o1, o2 = e1.vector_reset_tensor().clone(), e2.vector_reset_tensor().clone()
    assert torch.equal(o1, o2)
    out = pol.rollout(e1._dev, T, o1, want_logits=True)
    obs, prev, tr = o2, None, None
    for t in range(T):
        a = pol.forward(obs, prev, tr)
        np.testing.assert_allclose(out["actions"][t].cpu().numpy(), a.cpu().numpy(), atol=1e-6)
        wl, _ = P.rma_full(w, obs.cpu().numpy(), np.zeros((200, 4)) if prev is None else prev.cpu().numpy() * (1 - tr.cpu().numpy()[:, None]))
        np.testing.assert_allclose(out["logits"][t].cpu().numpy(), wl, atol=3e-5)
        ob, rw, trn = e2.vector_step_tensor(a)
        obs, prev, tr = ob.clone(), a, trn.clone()
        np.testing.assert_allclose(out["obs"][t].cpu().numpy(), obs.cpu().numpy(), atol=1e-5)
        assert torch.equal(out["truncated"][t], tr)
    assert int(out["truncated"].sum()) == 3 * 200      # max_steps = 9: every env truncated three times in 30 steps
    with pytest.raises(ValueError):
        pol.rollout(e1._dev, 2, torch.zeros((200, 21), device="cuda"))


def test_policy_nonstandard_sizes_use_the_interpreter(PG):
    """a network whose sizes differ from the training scripts' (param_embed_dim 5, 20 states) has no specialisation: the
    library falls back to the interpreter, same results as the float64 oracle"""
    from mujoco_drone_amd.policy import DevicePolicy
    from oracle import policy_ref as P
    rng = np.random.default_rng(4)
    ns, npar, na, emb, D = 20, 6, 4, 5, 26

    def fc(o, i):
        return (rng.normal(size=(o, i)) / np.sqrt(i)).astype(np.float32), (0.1 * rng.normal(size=o)).astype(np.float32)
    w = {}
    for name, (o, i) in {"param_encoder.0": (32, npar), "param_encoder.1": (emb, 32), "_hidden_layers.0": (256, ns + na + emb),
                         "_hidden_layers.1": (128, 256), "_logits.0": (128, 128), "_logits.1": (8, 128),
                         "_value_branch.0": (128, 128), "_value_branch.1": (128, 128), "_value_branch.2": (1, 128)}.items():
        w[name + "._model.0.weight"], w[name + "._model.0.bias"] = fc(o, i)
    w["_hidden_layers.2.weight"] = rng.uniform(0.5, 1.5, 128).astype(np.float32)
    w["_hidden_layers.2.bias"] = (0.1 * rng.normal(size=128)).astype(np.float32)
    w["_hidden_layers.2.running_mean"] = (0.2 * rng.normal(size=128)).astype(np.float32)
    w["_hidden_layers.2.running_var"] = rng.uniform(0.25, 1.75, 128).astype(np.float32)
    pol = DevicePolicy("RMA_full", w, obs_dim=D, num_states=ns, num_params=npar, num_actions=na)
    assert pol.kernel == 0
    n = 77
    obs = rng.normal(size=(n, D)).astype(np.float32); prev = rng.uniform(0, 1, (n, na)).astype(np.float32)
    act, logits, value = pol.forward(torch.tensor(obs, device="cuda"), torch.tensor(prev, device="cuda"), want_logits=True, want_value=True)
    wl, wv = P.rma_full(w, obs, prev, num_states=ns, num_params=npar)
    np.testing.assert_allclose(logits.cpu().numpy(), wl, atol=3e-5)
    np.testing.assert_allclose(value.cpu().numpy(), wv, atol=3e-5)
    np.testing.assert_allclose(act.cpu().numpy(), P.beta_mean_action(wl), atol=1e-5)


def test_policy_logp_and_beta_sampling(PG, kernel):
    """qd_policy_act: MyBetaDist.logp of the action it returns (deterministic and sampled) against the oracle's logp (pinned
    by the reference's MyBetaDist), and the sampled actions' distribution against Beta(alpha, beta): moments, a
    Kolmogorov-Smirnov distance per action dimension, reproducibility per (seed, counter)."""
    from scipy import stats
    from mujoco_drone_amd.policy import DevicePolicy
    from oracle import policy_ref as P
    w = weights_of(PG, "rma_full")
    pol = DevicePolicy("RMA_full", w)
    obs, prev = torch.tensor(PG["obs"], device="cuda"), torch.tensor(PG["prev_actions"], device="cuda")
    act, logp, logits = pol.forward(obs, prev, want_logp=True, want_logits=True)
    np.testing.assert_allclose(act.cpu().numpy(), PG["rma_full_action"], atol=1e-5)
    np.testing.assert_allclose(logp.cpu().numpy(), PG["rma_full_logp"], atol=2e-4)          # the reference's own logp
    # sampling: one observation repeated, so every row draws from the same four Beta distributions
    n = 16384
    o1, p1 = obs[3:4].repeat(n, 1).contiguous(), prev[3:4].repeat(n, 1).contiguous()
    a, lp, lg = pol.forward(o1, p1, explore=True, seed=11, counter=5, want_logp=True, want_logits=True)
    a, lp, lg = a.cpu().numpy().astype(np.float64), lp.cpu().numpy(), lg.cpu().numpy().astype(np.float64)
    assert a.min() > 0.0 and a.max() < 1.0
    np.testing.assert_allclose(lp, P.beta_logp(lg, a), atol=5e-4)
    al, be = P.beta_params(lg[0])
    mean, var = al / (al + be), al * be / ((al + be) ** 2 * (al + be + 1))
    np.testing.assert_allclose(a.mean(0), mean, atol=5 * np.sqrt(var / n).max())
    np.testing.assert_allclose(a.var(0), var, rtol=0.06)
    for d in range(4):
        ks = stats.kstest(a[:, d], stats.beta(al[d], be[d]).cdf).statistic
        assert ks < 1.63 / np.sqrt(n), (d, ks)                                           # 1 % level
    assert abs(np.corrcoef(a[:-1, 0], a[1:, 0])[0, 1]) < 0.03 and abs(np.corrcoef(a[:, 0], a[:, 1])[0, 1]) < 0.03
    a2 = pol.forward(o1, p1, explore=True, seed=11, counter=5).cpu().numpy()
    np.testing.assert_array_equal(a2, a.astype(np.float32))                             # same stream -> same draw
    a3 = pol.forward(o1, p1, explore=True, seed=11, counter=6).cpu().numpy()
    a4 = pol.forward(o1, p1, explore=True, seed=12, counter=5).cpu().numpy()
    assert np.mean(a3 == a2) < 0.01 and np.mean(a4 == a2) < 0.01


def test_policy_exploring_rollout_is_a_sample_batch(PG, kernel):
    """qd_rollout_policy with explore: step t draws from stream (seed, counter0 + t); logp / value / logits columns equal
    what T x (qd_policy_act, qd_step) produce"""
    from mujoco_drone_amd.policy import DevicePolicy
    from mujoco_drone_amd.environments.BaseDroneEnv import base_config
    from mujoco_drone_amd.environments.observation_wrappers import LocalFrameRPYParamsEnv
    from mujoco_drone_amd.environments.rewards import distance_energy_reward
    pol = DevicePolicy("RMA_full", weights_of(PG, "rma_full"))
    cfg = dict(base_config, num_drones=100, reward_fcn=distance_energy_reward, random_params=True, param_difficulty=1,
               state_difficulty=0.2, max_steps=1024, auto_reset=True)
    T = 12
    e1, e2 = LocalFrameRPYParamsEnv(cfg), LocalFrameRPYParamsEnv(cfg)
    o1, o2 = e1.vector_reset_tensor().clone(), e2.vector_reset_tensor().clone()
    out = pol.rollout(e1._dev, T, o1, explore=True, seed=99, counter0=40, want_logp=True, want_value=True)
    obs, prev, tr = o2, None, None
    for t in range(T):
        a, lp, v = pol.forward(obs, prev, tr, explore=True, seed=99, counter=40 + t, want_logp=True, want_value=True)
        np.testing.assert_allclose(out["actions"][t].cpu().numpy(), a.cpu().numpy(), atol=1e-6)
        np.testing.assert_allclose(out["logp"][t].cpu().numpy(), lp.cpu().numpy(), atol=1e-4)
        np.testing.assert_allclose(out["value"][t].cpu().numpy(), v.cpu().numpy(), atol=1e-5)
        ob, rw, trn = e2.vector_step_tensor(a)
        obs, prev, tr = ob.clone(), a, trn.clone()
        np.testing.assert_allclose(out["reward"][t].cpu().numpy(), rw.cpu().numpy(), atol=1e-5)
    assert float(out["actions"].std()) > 0.05


@pytest.mark.parametrize("tile", [16, 32])
def test_fused_rollout_equals_two_launch_loop(PG, monkeypatch, tile):
    """k_rollout_fused_pipe (one launch per fragment, the env step beside the forward pass) against the per-step path driven from Python (qd_policy_act + qd_step),
    deterministic and exploring, with in-kernel auto-resets (truncation every 9 steps) and a ragged env count; both copies of the
    kernel (16 and 32 envs per workgroup: qd_rollout_fused.hip / qd_rollout_fused32.hip)"""
    monkeypatch.setenv("QD_FUSED_TILE", str(tile))
    from mujoco_drone_amd.policy import DevicePolicy
    from mujoco_drone_amd.environments.BaseDroneEnv import base_config
    from mujoco_drone_amd.environments.observation_wrappers import LocalFrameRPYParamsEnv
    from mujoco_drone_amd.environments.rewards import distance_energy_reward
    monkeypatch.delenv("QD_POLICY_GENERIC", raising=False)
    for tag, fam in dict(TAGS, custom_mlp="CustomMLP", rma_smaller="RMA_model_smaller").items():   # every network with a fused instantiation
        pol = DevicePolicy(fam, weights_of(PG, tag))
        assert pol.kernel > 0
        cfg = dict(base_config, num_drones=203, reward_fcn=distance_energy_reward, random_params=True, param_difficulty=1,
                   state_difficulty=0.2, max_steps=9, auto_reset=True)
        for explore in (False, True):
            T = 25
            e1, e2 = LocalFrameRPYParamsEnv(cfg), LocalFrameRPYParamsEnv(cfg)
            o1, o2 = e1.vector_reset_tensor().clone(), e2.vector_reset_tensor().clone()
            out = pol.rollout(e1._dev, T, o1, explore=explore, seed=5, counter0=7, want_logp=True, want_value=True, want_logits=True)
            obs, prev, tr = o2, None, None
            for t in range(T):
                a, lp, lg, v = pol.forward(obs, prev, tr, explore=explore, seed=5, counter=7 + t, want_logp=True, want_logits=True, want_value=True)
                np.testing.assert_allclose(out["logits"][t].cpu().numpy(), lg.cpu().numpy(), atol=2e-5, err_msg="%s t=%d" % (tag, t))
                np.testing.assert_allclose(out["actions"][t].cpu().numpy(), a.cpu().numpy(), atol=2e-5)
                np.testing.assert_allclose(out["logp"][t].cpu().numpy(), lp.cpu().numpy(), atol=2e-3)
                np.testing.assert_allclose(out["value"][t].cpu().numpy(), v.cpu().numpy(), atol=2e-5)
                ob, rw, trn = e2.vector_step_tensor(out["actions"][t])      # same actions into the reference env: no drift
                obs, prev, tr = ob.clone(), out["actions"][t], trn.clone()
                np.testing.assert_allclose(out["obs"][t].cpu().numpy(), obs.cpu().numpy(), atol=2e-5)
                np.testing.assert_allclose(out["reward"][t].cpu().numpy(), rw.cpu().numpy(), atol=2e-5)
                assert torch.equal(out["truncated"][t], tr)
            # the row wave's bounded poll for the solver wave's publication never ran out (qd_health_counters)
            assert e1._dev.health_counters() == (0,)
            assert int(out["truncated"].sum()) == 2 * 203
            q1 = [x.cpu().numpy() for x in e1._dev.get_state()]
            q2 = [x.cpu().numpy() for x in e2._dev.get_state()]
            for x, y in zip(q1, q2):
                np.testing.assert_allclose(x, y, atol=2e-5)


@pytest.mark.parametrize("n", [300, 4096 + 160])
def test_fused_closed_loop_vs_the_float64_oracles(PG, orc, monkeypatch, n):
    """The closed policy -> env loop of ONE launch (k_rollout_fused_pipe, 16 and 32 envs per workgroup) against nothing of the HIP
    library: the float64 restatement of RMA_full (oracle/policy_ref.py, pinned by the reference classes' own outputs) in closed loop
    with the float64 restatement of the env step (oracle/qd_oracle.c), 40 steps from the same random states and domain-randomised
    parameters.  The loop feeds differences back through the policy; the bar on the state at the end is BASELINE's 1e-4 of the open
    loop, rows / actions / rewards on the way are held to a few times their measured maxima (printed)."""
    from mujoco_drone_amd import _lib as L
    from mujoco_drone_amd.environments import _device
    from mujoco_drone_amd.policy import DevicePolicy
    from oracle import policy_ref as P
    from test_gpu_parity import make_cfg
    from divergence import Divergence
    monkeypatch.delenv("QD_POLICY_GENERIC", raising=False)
    monkeypatch.delenv("QD_FUSED_TILE", raising=False)
    w = weights_of(PG, "rma_full")
    pol = DevicePolicy("RMA_full", w)
    assert pol.kernel > 0
    T = 40
    c = make_cfg(L, n, load=True, start=1, random_params=1, seed=11, difficulty=1.0, sdiff=0.2, max_steps=10 ** 6, max_distance=1e9)
    env = _device.DeviceEnv(c)
    o0 = env.reset().clone()
    raw = env.get_params().cpu().numpy()
    q0, v0, a0, _, _ = [x.cpu().numpy().astype(np.float64) for x in env.get_state()]
    ob = orc.Batch(raw, True, L.OBS_KINDS.index("LocalFrameRPYParamsEnv"), L.REWARD_KINDS.index("distance_energy_reward"),
                   0.01, 1, 1, (0, 0, 15, 0), 1e9, 10 ** 6)
    ob.qpos[:], ob.qvel[:], ob.act[:] = q0, v0, a0
    out = pol.rollout(env, T, o0, want_logits=True, want_value=True)
    obs, prev = o0.cpu().numpy().astype(np.float64), np.zeros((n, 4))
    worst = dict(obs=0.0, act=0.0, rew=0.0, val=0.0)
    for t in range(T):
        lg, val = P.rma_full(w, obs, prev)
        a = P.beta_mean_action(lg)
        worst["act"] = max(worst["act"], float(np.abs(out["actions"][t].cpu().numpy() - a).max()))
        worst["val"] = max(worst["val"], float(np.abs(out["value"][t].cpu().numpy() - val).max()))
        oo, rr, tr = ob.step(a, threads=8)
        d = np.abs(out["obs"][t].cpu().numpy().astype(np.float64) - oo)
        d[:, 5] = np.minimum(d[:, 5], np.abs(d[:, 5] - 2 * np.pi))
        worst["obs"] = max(worst["obs"], float(d.max()))
        worst["rew"] = max(worst["rew"], float(np.abs(out["reward"][t].cpu().numpy() - rr).max()))
        assert not tr.any() and int(out["truncated"][t].sum()) == 0
        obs, prev = oo.copy(), a
    gq, gv, ga, _, gk = [x.cpu().numpy().astype(np.float64) for x in env.get_state()]
    assert np.all(gk == T)
    div = Divergence(True)
    div.update(dict(qpos=gq, qvel=gv, act=ga), dict(qpos=ob.qpos, qvel=ob.qvel, act=ob.act))
    print(div.table("closed loop RMA_full, %d envs, step %d: fused kernel vs float64 oracles" % (n, T)))
    print("max over %d steps: |obs| %.3e |action| %.3e |reward| %.3e |value| %.3e" % (T, worst["obs"], worst["act"], worst["rew"], worst["val"]))
    assert div.max("rel") < 1e-4 and div.max("mixed") < 1e-4      # BASELINE's bar; measured on MI355X: 1.8e-6
    # ~8 x the measured maxima (6.4e-6, 2.1e-7, 4.8e-6, 7.6e-8): the float16-pair layers are as close to float64 as float32 ones
    assert worst["obs"] < 5e-5 and worst["act"] < 2e-6 and worst["rew"] < 4e-5 and worst["val"] < 1e-6


@pytest.mark.parametrize("tile", [16, 0])
def test_pipelined_rollout_many_workgroups_distance_truncation_moving_reference(PG, monkeypatch, tile):
    """k_rollout_fused_pipe (env step beside the forward pass) past one workgroup per CU: 4096 + 37 envs = 259 workgroups of 16 (forced)
    or, as the library chooses above 4096 envs, 130 of 32, the last one ragged; truncation by distance as well as by step count
    (max_distance 0.35 at state_difficulty 1), a circling waypoint, and the arena left as the per-step path leaves it.
    Reference: qd_policy_act + qd_step per step on a twin env fed the same actions."""
    if tile:
        monkeypatch.setenv("QD_FUSED_TILE", str(tile))
    else:
        monkeypatch.delenv("QD_FUSED_TILE", raising=False)
    from mujoco_drone_amd.policy import DevicePolicy
    from mujoco_drone_amd.environments.BaseDroneEnv import base_config
    from mujoco_drone_amd.environments.observation_wrappers import LocalFrameRPYParamsEnv
    from mujoco_drone_amd.environments.rewards import distance_energy_reward
    monkeypatch.delenv("QD_POLICY_GENERIC", raising=False)
    pol = DevicePolicy("RMA_full", weights_of(PG, "rma_full"))
    assert pol.kernel > 0
    n, T = 4096 + 37, 14
    cfg = dict(base_config, num_drones=n, reward_fcn=distance_energy_reward, random_params=True, param_difficulty=1,
               state_difficulty=1.0, max_steps=11, max_distance=0.35, auto_reset=True,
               reference_trajectory={"type": "circle", "radius": 0.3, "frequency": 0.5})
    e1, e2 = LocalFrameRPYParamsEnv(cfg), LocalFrameRPYParamsEnv(cfg)
    o1, o2 = e1.vector_reset_tensor().clone(), e2.vector_reset_tensor().clone()
    assert torch.equal(o1, o2)
    out = pol.rollout(e1._dev, T, o1, want_logits=True)
    obs, prev, tr = o2, None, None
    by_distance = 0
    for t in range(T):
        a, lg = pol.forward(obs, prev, tr, want_logits=True)
        np.testing.assert_allclose(out["logits"][t].cpu().numpy(), lg.cpu().numpy(), atol=2e-5, err_msg="t=%d" % t)
        np.testing.assert_allclose(out["actions"][t].cpu().numpy(), a.cpu().numpy(), atol=2e-5)
        ob, rw, trn = e2.vector_step_tensor(out["actions"][t])
        obs, prev, tr = ob.clone(), out["actions"][t], trn.clone()
        np.testing.assert_allclose(out["obs"][t].cpu().numpy(), obs.cpu().numpy(), atol=2e-5, err_msg="t=%d" % t)
        np.testing.assert_allclose(out["reward"][t].cpu().numpy(), rw.cpu().numpy(), atol=2e-5)
        assert torch.equal(out["truncated"][t], tr)
        if t < 10:
            by_distance += int(tr.sum())
    assert by_distance > n // 20                      # the distance rule fired well before max_steps did
    for x, y in zip(e1._dev.get_state(), e2._dev.get_state()):
        np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), atol=2e-5)
    # the next per-step launch continues from what the fragment left (stale sensor flags, episode counters, activations)
    act = torch.rand((n, 4), device="cuda")
    oa, ra, ta = e1.vector_step_tensor(act)
    ob, rb, tb = e2.vector_step_tensor(act)
    np.testing.assert_allclose(oa.cpu().numpy(), ob.cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(ra.cpu().numpy(), rb.cpu().numpy(), atol=2e-5)
    assert torch.equal(ta, tb)


def test_pipelined_rollout_config5_full_size(PG, monkeypatch):
    """BASELINE config 5 (train_LSTM.py: 8192 envs, LocalFrameFullStateEnv rows with the accelerometer AND the activations, circling
    waypoint) through the pipelined kernel: what a row carries of the action -- the four activations, and a reset lane's reading at
    its new state -- is written a pass late; 8192 + 5 envs = 513 workgroups, resets every 6 steps."""
    from mujoco_drone_amd.policy import DevicePolicy
    from mujoco_drone_amd.environments.BaseDroneEnv import base_config
    from mujoco_drone_amd.environments.observation_wrappers import LocalFrameFullStateEnv
    from mujoco_drone_amd.environments.rewards import distance_energy_reward_pendulum_en4
    monkeypatch.delenv("QD_POLICY_GENERIC", raising=False)
    pol = DevicePolicy("CNNestimator", weights_of(PG, "cnn_est_ff"), obs_dim=23, num_states=23)
    assert pol.kernel > 0
    n, T = 8192 + 5, 15
    cfg = dict(base_config, num_drones=n, reward_fcn=distance_energy_reward_pendulum_en4, random_params=False, state_difficulty=0.8,
               max_steps=6, auto_reset=True, reference_trajectory=dict(type="circle", radius=1.0, frequency=0.5))
    e1, e2 = LocalFrameFullStateEnv(cfg), LocalFrameFullStateEnv(cfg)
    o1, o2 = e1.vector_reset_tensor().clone(), e2.vector_reset_tensor().clone()
    out = pol.rollout(e1._dev, T, o1, want_logits=True, want_value=True)
    obs, prev, tr = o2, None, None
    for t in range(T):
        a, lg, v = pol.forward(obs, prev, tr, want_logits=True, want_value=True)
        np.testing.assert_allclose(out["logits"][t].cpu().numpy(), lg.cpu().numpy(), atol=2e-5, err_msg="t=%d" % t)
        np.testing.assert_allclose(out["value"][t].cpu().numpy(), v.cpu().numpy(), atol=2e-5)
        ob, rw, trn = e2.vector_step_tensor(out["actions"][t])
        obs, prev, tr = ob.clone(), out["actions"][t], trn.clone()
        d = (out["obs"][t] - obs).abs()
        assert float(d[:, 12:15].max()) <= 2e-3, (t, float(d[:, 12:15].max()))     # the accelerometer (tens of m/s^2, differences of forces)
        d[:, 12:15] = 0
        assert float(d.max()) <= 2e-5, (t, float(d.max()), int(d.argmax()) % 23)
        np.testing.assert_allclose(out["reward"][t].cpu().numpy(), rw.cpu().numpy(), atol=5e-5)
        assert torch.equal(out["truncated"][t], tr)
    assert int(out["truncated"].sum()) == 2 * n
    for x, y in zip(e1._dev.get_state(), e2._dev.get_state()):
        np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), atol=2e-5)
    act = torch.rand((n, 4), device="cuda")          # the arena the fragment left: readings, flags, counters
    oa, ra, ta = e1.vector_step_tensor(act)
    ob, rb, tb = e2.vector_step_tensor(act)
    d = (oa - ob).abs()
    assert float(d[:, 12:15].max()) <= 2e-3
    d[:, 12:15] = 0
    assert float(d.max()) <= 2e-5 and float((ra - rb).abs().max()) <= 5e-5 and torch.equal(ta, tb)


HIST = {"RMA_full_adapt": ("rma_adapt", 22, lambda P, w, oh, ah: P.rma_full_adapt(w, oh, ah)[:2]),
        "CNNestimator_estimate": ("cnn_est_hist", 23, lambda P, w, oh, ah: P.cnn_estimator_hist(w, oh, ah)[:2])}


@pytest.mark.parametrize("family", list(HIST))
def test_adaptation_policy_incremental_history_vs_full_recomputation(PG, kernel, family):
    """RMA_full with the adaptation CNN (train_RMA.py's configuration): the device evaluates it incrementally from per-env
    rings, one new inMLP / conv1 / conv2 value per step; the oracle re-runs the whole TimeCNN2 on the explicit 32-step
    zero-padded window every step, as the reference does.  50 steps, episodes restarting at different times (history
    re-initialised inside the kernel from the prev_truncated flags), both step parities, ragged batch."""
    from mujoco_drone_amd.policy import DevicePolicy
    from oracle import policy_ref as P
    rng = np.random.default_rng(12)
    tag, D, full = HIST[family]
    w = weights_of(PG, tag)
    pol = DevicePolicy(family, w, obs_dim=D, num_states=23 if D == 23 else 16)
    assert (pol.kernel > 0) == (kernel == "specialised") and pol.has_history
    n, T, Lw = 37, 50, 32
    obs_seq = rng.normal(scale=1.2, size=(T, n, D)).astype(np.float32)
    act_seq = rng.uniform(0, 1, (T, n, 4)).astype(np.float32)          # the action taken AFTER obs_seq[t] (fed back as previous action)
    start = np.zeros(n, dtype=np.int64)                                  # first step of each env's current episode
    restart = {7: np.arange(n) % 3 == 0, 20: np.arange(n) % 2 == 1, 21: np.arange(n) % 5 == 0, 45: np.ones(n, bool)}
    pol.reset_state(n)
    worst = 0.0
    for t in range(T):
        fresh = restart.get(t, np.zeros(n, bool)) if t > 0 else np.zeros(n, bool)
        start[fresh] = t
        prev = np.where((start == t)[:, None], 0.0, act_seq[t - 1] if t > 0 else np.zeros((n, 4))).astype(np.float32)
        a, logits, value = pol.forward(torch.tensor(obs_seq[t], device="cuda"), torch.tensor(prev, device="cuda"),
                                       torch.tensor(fresh.astype(np.uint8), device="cuda") if t > 0 else None,
                                       counter=t, want_logits=True, want_value=True)
        # the explicit windows the reference's view requirements would hand to the model
        oh, ah = np.zeros((n, Lw, D)), np.zeros((n, Lw, 4))
        for j in range(Lw):
            tau = t - (Lw - 1) + j
            live = tau >= start
            if tau >= 0:
                oh[live, j] = obs_seq[tau][live]
                if tau >= 1:
                    inside = live & (tau - 1 >= start)                  # the action before the episode's first obs is zero
                    ah[inside, j] = act_seq[tau - 1][inside]
        wl, wv = full(P, w, oh, ah)
        worst = max(worst, float(np.abs(logits.cpu().numpy() - wl).max()), float(np.abs(value.cpu().numpy() - wv).max()))
        np.testing.assert_allclose(logits.cpu().numpy(), wl, atol=3e-5, err_msg="t=%d" % t)
        np.testing.assert_allclose(value.cpu().numpy(), wv, atol=3e-5, err_msg="t=%d" % t)
        np.testing.assert_allclose(a.cpu().numpy(), P.beta_mean_action(wl), atol=1e-5)
    print("%s, incremental vs full recomputation: worst |logit / value error| %.2e" % (family, worst))
    with pytest.raises(ValueError):
        pol.lib and __import__("mujoco_drone_amd._lib", fromlist=["check"]).check(
            pol.lib.qd_policy_forward(pol.handle, n, None, None, None, None, None, None, None))


def test_adaptation_policy_rollout(PG, kernel):
    """closed loop with the windowed policy: qd_rollout_policy carries the history through the fragment (and restarts it
    for envs the step kernel re-sampled) exactly as T x (qd_policy_act, qd_step) does"""
    from mujoco_drone_amd.policy import DevicePolicy
    from mujoco_drone_amd.environments.BaseDroneEnv import base_config
    from mujoco_drone_amd.environments.observation_wrappers import LocalFrameRPYParamsEnv
    from mujoco_drone_amd.environments.rewards import distance_energy_reward
    w = weights_of(PG, "rma_adapt")
    p1, p2 = DevicePolicy("RMA_full_adapt", w), DevicePolicy("RMA_full_adapt", w)
    cfg = dict(base_config, num_drones=150, reward_fcn=distance_energy_reward, random_params=True, param_difficulty=1,
               state_difficulty=0.2, max_steps=11, auto_reset=True)
    T = 40
    e1, e2 = LocalFrameRPYParamsEnv(cfg), LocalFrameRPYParamsEnv(cfg)
    o1, o2 = e1.vector_reset_tensor().clone(), e2.vector_reset_tensor().clone()
    p1.reset_state(150); p2.reset_state(150)
    out = p1.rollout(e1._dev, T, o1, counter0=3, want_logits=True, want_value=True)
    obs, prev, tr = o2, None, None
    for t in range(T):
        a, lg, v = p2.forward(obs, prev, tr, counter=3 + t, want_logits=True, want_value=True)
        np.testing.assert_allclose(out["logits"][t].cpu().numpy(), lg.cpu().numpy(), atol=1e-5, err_msg="t=%d" % t)
        np.testing.assert_allclose(out["value"][t].cpu().numpy(), v.cpu().numpy(), atol=1e-5)
        ob, rw, trn = e2.vector_step_tensor(out["actions"][t])
        obs, prev, tr = ob.clone(), out["actions"][t], trn.clone()
        assert torch.equal(out["truncated"][t], tr)
    assert int(out["truncated"].sum()) == 3 * 150
    np.testing.assert_array_equal(p1.state.cpu().numpy(), p2.state.cpu().numpy())


def test_cnn_estimator_forward_and_fused_config5_rollout(PG, kernel):
    """train_LSTM.py's network (CNNestimator, feed-forward mode) against the reference model's outputs, and closed-loop on the
    BASELINE config-5 env (LocalFrameFullStateEnv, pendulum-energy reward, circling waypoint): the fused one-launch rollout
    equals the per-step loop"""
    from mujoco_drone_amd.policy import DevicePolicy
    from mujoco_drone_amd.environments.BaseDroneEnv import base_config
    from mujoco_drone_amd.environments.observation_wrappers import LocalFrameFullStateEnv
    from mujoco_drone_amd.environments.rewards import distance_energy_reward_pendulum_en4
    pol = DevicePolicy("CNNestimator", weights_of(PG, "cnn_est_ff"), obs_dim=23, num_states=23)
    assert (pol.kernel > 0) == (kernel == "specialised")
    act, logits, value = pol.forward(torch.tensor(PG["obs23"], device="cuda"), torch.tensor(PG["prev_actions"], device="cuda"),
                                     want_logits=True, want_value=True)
    np.testing.assert_allclose(logits.cpu().numpy(), PG["cnn_est_ff_logits"], atol=2e-5)
    np.testing.assert_allclose(value.cpu().numpy(), PG["cnn_est_ff_value"], atol=2e-5)
    cfg = dict(base_config, num_drones=120, reward_fcn=distance_energy_reward_pendulum_en4, random_params=False, state_difficulty=0.8,
               max_steps=13, auto_reset=True, reference_trajectory=dict(type="circle", radius=1.0, frequency=0.5))
    T = 30
    e1, e2 = LocalFrameFullStateEnv(cfg), LocalFrameFullStateEnv(cfg)
    o1, o2 = e1.vector_reset_tensor().clone(), e2.vector_reset_tensor().clone()
    out = pol.rollout(e1._dev, T, o1, want_logits=True)
    obs, prev, tr = o2, None, None
    for t in range(T):
        a, lg = pol.forward(obs, prev, tr, want_logits=True)
        np.testing.assert_allclose(out["logits"][t].cpu().numpy(), lg.cpu().numpy(), atol=2e-5, err_msg="t=%d" % t)
        ob, rw, trn = e2.vector_step_tensor(out["actions"][t])
        obs, prev, tr = ob.clone(), out["actions"][t], trn.clone()
        np.testing.assert_allclose(out["obs"][t].cpu().numpy(), obs.cpu().numpy(), atol=2e-5)
        np.testing.assert_allclose(out["reward"][t].cpu().numpy(), rw.cpu().numpy(), atol=5e-5)
        assert torch.equal(out["truncated"][t], tr)
    assert int(out["truncated"].sum()) >= 2 * 120


def test_dataset_collection_like_rollout_py(PG, tmp_path):
    """rollout.py:64-86 on the GPU: batches {'z','o','a','t'}; z = policy.model.z = param_encoder(env parameters) against the
    oracle; parameters are regenerated per batch; pickle round trip"""
    import pickle
    from mujoco_drone_amd.policy import DevicePolicy
    from mujoco_drone_amd.rollout import collect_dataset, write_dataset
    from mujoco_drone_amd.environments.BaseDroneEnv import base_config
    from mujoco_drone_amd.environments.observation_wrappers import LocalFrameRPYParamsEnv
    from mujoco_drone_amd.environments.rewards import distance_energy_reward
    from oracle import policy_ref as P
    w = weights_of(PG, "rma_full")
    pol = DevicePolicy("RMA_full", w)
    env = LocalFrameRPYParamsEnv(dict(base_config, num_drones=64, reward_fcn=distance_energy_reward, random_params=True, param_difficulty=1,
                                      state_difficulty=0.3, max_steps=4096, max_distance=3, auto_reset=True))
    env.vector_reset_tensor()
    batches = collect_dataset(env, pol, num_batches=3, rollout_length=6, explore=True, seed=4)
    assert len(batches) == 3
    for b in batches:
        assert b["z"].shape == (64, 8) and b["o"].shape == (6, 64, 22) and b["a"].shape == (6, 64, 4) and b["t"].shape == (6, 64)
        want = P._seq(w, "param_encoder", b["o"][-1][:, -6:].astype(np.float64), ["tanh", None])   # the env's parameters -> z
        np.testing.assert_allclose(b["z"], want, atol=1e-5)
        assert b["a"].min() > 0 and b["a"].max() < 1 and b["t"].dtype == bool
    assert np.abs(batches[0]["z"] - batches[1]["z"]).max() > 1e-3      # regen: new parameters, new embedding
    write_dataset(str(tmp_path / "dataset.pickle"), batches)
    back = pickle.load(open(tmp_path / "dataset.pickle", "rb"))
    np.testing.assert_array_equal(back[2]["o"], batches[2]["o"])
    nested = collect_dataset(env, pol, 1, 4, as_lists=True)
    assert isinstance(nested[0]["o"], list) and len(nested[0]["o"]) == 4 and nested[0]["o"][0].shape == (64, 22)


def test_beta_head_at_extreme_logits(PG, kernel):
    """MyBetaDist at the edges of its clamp (logits from -60 to +60 -> alpha, beta in [1, 51]): mean action, log-prob of the mean
    and of samples against the float64 oracle (lgamma by Stirling on the device)"""
    from mujoco_drone_amd.policy import DevicePolicy
    from oracle import policy_ref as P
    w = dict(weights_of(PG, "rma_full"))
    vals = np.array([-60.0, -50.0, -5.0, 0.0, 5.0, 20.0, 50.0, 60.0], np.float32)
    rng = np.random.default_rng(0)
    n = 512
    want_logits = vals[rng.integers(0, len(vals), size=(n, 8))]
    # the last actor layer outputs its bias only; one network per row would be wasteful, so rows share 8 bias patterns
    patterns = want_logits[:8]
    for k in range(8):
        w["_logits.1._model.0.weight"] = np.zeros_like(w["_logits.1._model.0.weight"])
        w["_logits.1._model.0.bias"] = patterns[k].copy()
        pol = DevicePolicy("RMA_full", w)
        obs = torch.tensor(PG["obs"][:16], device="cuda")
        a, lp, lg = pol.forward(obs, want_logp=True, want_logits=True)
        lg64 = lg.cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(lg64, np.tile(patterns[k], (16, 1)), atol=1e-6)
        np.testing.assert_allclose(a.cpu().numpy(), P.beta_mean_action(lg64), rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(lp.cpu().numpy(), P.beta_logp(lg64, a.cpu().numpy().astype(np.float64)), rtol=2e-5, atol=3e-4)
        o2 = obs.repeat(64, 1).contiguous()
        s, lps = pol.forward(o2, explore=True, seed=k, counter=1, want_logp=True)
        s64 = s.cpu().numpy().astype(np.float64)
        assert np.all(np.isfinite(s64)) and s64.min() >= 0.0 and s64.max() <= 1.0
        np.testing.assert_allclose(lps.cpu().numpy(), P.beta_logp(np.tile(patterns[k], (len(s64), 1)).astype(np.float64), s64), rtol=3e-5, atol=1e-3)
        al, be = P.beta_params(patterns[k].astype(np.float64))
        np.testing.assert_allclose(s64.mean(0), al / (al + be), atol=0.03)


def test_policy_api_error_paths(PG):
    """the C ABI refuses misuse with a status and a message instead of launching"""
    import ctypes as C
    from mujoco_drone_amd import _lib as L
    from mujoco_drone_amd.policy import DevicePolicy, compile_program
    lib = L.lib()
    w = weights_of(PG, "rma_full")
    pol = DevicePolicy("RMA_full", w)
    obs = torch.zeros((8, 22), device="cuda")
    out = torch.zeros((8, 4), device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())
    assert lib.qd_policy_forward(pol.handle, 8, P(obs), None, None, None, None, None, None) != 0 and "no output" in L.last_error()
    assert lib.qd_policy_forward(pol.handle, 0, P(obs), None, None, P(out), None, None, None) != 0
    assert lib.qd_policy_forward(None, 8, P(obs), None, None, P(out), None, None, None) != 0 and "null policy" in L.last_error()
    assert lib.qd_policy_act(pol.handle, 8, None, None, None, 0, 0, 0, None, P(out), None, None, None, None) != 0
    # packed buffer too small / misaligned
    d, ops, blob = compile_program("RMA_full", w)
    nbytes = lib.qd_policy_packed_bytes(C.byref(d), ops)
    small = torch.empty(nbytes - 16, dtype=torch.uint8, device="cuda")
    h = C.c_void_p()
    assert lib.qd_policy_create(C.byref(d), ops, blob.ctypes.data_as(C.c_void_p), blob.size, P(small), nbytes - 16, C.byref(h)) == L.QD_ERR_ARENA
    # a windowed policy needs its state buffer and the step counter
    wa = weights_of(PG, "rma_adapt")
    pa = DevicePolicy("RMA_full_adapt", wa)
    assert lib.qd_policy_forward(pa.handle, 8, P(obs), None, None, P(out), None, None, None) != 0 and "qd_policy_act" in L.last_error()
    assert lib.qd_policy_act(pa.handle, 8, P(obs), None, None, 0, 0, 0, None, P(out), None, None, None, None) != 0 and "state" in L.last_error()
    aux = torch.zeros((8, 8), device="cuda")
    assert lib.qd_policy_aux(pa.handle, 8, P(obs), None, None, P(aux), None) == L.QD_ERR_UNSUPPORTED
    assert lib.qd_policy_state_bytes(pol.handle, 100) == 0 and lib.qd_policy_state_bytes(pa.handle, 100) == 100 * 704 * 4
    # obs width mismatch between env and policy in a rollout
    from mujoco_drone_amd.environments.BaseDroneEnv import base_config
    from mujoco_drone_amd.environments.observation_wrappers import LocalFrameFullStateEnv
    env = LocalFrameFullStateEnv(dict(base_config, num_drones=16))
    o = env.vector_reset_tensor()
    with pytest.raises(ValueError):
        pol.rollout(env._dev, 2, o)


def test_lstm_estimator_policy_vs_reference_model(PG, kernel):
    """LSTMestimator with the nn.LSTM estimate in the loop: the device steps it one observation at a time (h, c and the
    previous observation in per-env rings) and must reproduce the reference model's own outputs over whole 24-step
    episodes; then episodes restarted in the middle against the float64 oracle"""
    from mujoco_drone_amd.policy import DevicePolicy
    from oracle import policy_ref as P
    w = weights_of(PG, "lstm_est")
    pol = DevicePolicy("LSTMestimator_estimate", w, obs_dim=19, num_states=19)
    assert (pol.kernel > 0) == (kernel == "specialised") and pol.has_history
    o, a = PG["lstm_est_obs_seq"], PG["lstm_est_action_seq"]
    Bn, Tn = o.shape[:2]
    pol.reset_state(Bn)
    for t in range(Tn):
        prev = torch.tensor(a[:, t - 1], device="cuda") if t > 0 else None
        act, logits, value = pol.forward(torch.tensor(o[:, t], device="cuda"), prev, None, counter=t, want_logits=True, want_value=True)
        np.testing.assert_allclose(logits.cpu().numpy(), PG["lstm_est_logits"][:, t], atol=3e-5, err_msg="t=%d" % t)
        np.testing.assert_allclose(value.cpu().numpy(), PG["lstm_est_value"][:, t], atol=3e-5)
    # restart every third env at step 9: its h, c, previous observation and previous action go back to zero
    rng = np.random.default_rng(2)
    n, T2 = 30, 20
    obs = rng.normal(size=(n, T2, 19)).astype(np.float32); acts = rng.uniform(0, 1, (n, T2, 4)).astype(np.float32)
    fresh9 = np.arange(n) % 3 == 0
    pol.reset_state(n)
    got = []
    for t in range(T2):
        fr = fresh9 if t == 9 else np.zeros(n, bool)
        prev = acts[:, t - 1] * (1 - fr[:, None]) if t > 0 else np.zeros((n, 4), np.float32)
        _, lg = pol.forward(torch.tensor(obs[:, t], device="cuda"), torch.tensor(prev.astype(np.float32), device="cuda"),
                            torch.tensor(fr.astype(np.uint8), device="cuda") if t > 0 else None, counter=100 + t, want_logits=True)
        got.append(lg.cpu().numpy())
    got = np.stack(got, 1)
    want_all, _, _ = P.lstm_estimator(w, obs, acts)
    want_restart, _, _ = P.lstm_estimator(w, obs[fresh9, 9:], acts[fresh9, 9:])
    np.testing.assert_allclose(got[~fresh9], want_all[~fresh9], atol=3e-5)
    np.testing.assert_allclose(got[fresh9, :9], want_all[fresh9, :9], atol=3e-5)
    np.testing.assert_allclose(got[fresh9, 9:], want_restart, atol=3e-5)
    # without the estimate the network is feed-forward on the 19-value observation
    ff = DevicePolicy("LSTMestimator", w, obs_dim=19, num_states=19)
    assert not ff.has_history
    _, lg = ff.forward(torch.tensor(obs[:, 3], device="cuda"), torch.tensor(acts[:, 2], device="cuda"), want_logits=True)
    want_ff, _, _ = P.lstm_estimator(w, obs[:, 2:4], acts[:, 2:4], use_estimate=False)
    np.testing.assert_allclose(lg.cpu().numpy(), want_ff[:, 1], atol=3e-5)


def test_evaluate_trajectory_like_evaluation_py(PG, golden_traj=None):
    """evaluation.py:38-72 as one device-side rollout: waypoint k is set before step k (the action for that step comes from the
    observation under the previous waypoint); against the same loop written out with env.reference = x per step"""
    from mujoco_drone_amd.policy import DevicePolicy
    from mujoco_drone_amd.evaluation import evaluate_trajectory, gen_step_trajectory, gen_ramp_trajectory, gen_circle_trajectory
    from mujoco_drone_amd.environments.BaseDroneEnv import base_config
    from mujoco_drone_amd.environments.observation_wrappers import LocalFrameRPYParamsEnv
    from mujoco_drone_amd.environments.rewards import distance_energy_reward
    G = np.load(os.path.join(HERE, "golden", "reference_vectors.npz"))
    _, st = gen_step_trajectory(G["traj_step_args"][0], G["traj_step_args"][1], G["traj_start"], G["traj_end"])
    _, rp = gen_ramp_trajectory(G["traj_ramp_args"][0], G["traj_ramp_args"][1], G["traj_start"], G["traj_end"])
    _, ci = gen_circle_trajectory(2.0, 0.5, 1.0, 15.0)
    np.testing.assert_array_equal(st, G["traj_step"]); np.testing.assert_allclose(rp, G["traj_ramp"], atol=1e-13)
    np.testing.assert_allclose(ci, G["traj_circle"], atol=1e-13)         # the generators equal the reference's outputs
    pol = DevicePolicy("RMA_full", weights_of(PG, "rma_full"))
    cfg = dict(base_config, num_drones=8, reward_fcn=distance_energy_reward, random_params=True, param_difficulty=1,
               state_difficulty=0.2, max_steps=2048, max_distance=1e9, controlled=True)
    traj = rp[:60]
    e1, e2 = LocalFrameRPYParamsEnv(cfg), LocalFrameRPYParamsEnv(cfg)
    observations, actions, rewards, out = evaluate_trajectory(e1, pol, traj)
    assert len(observations) == len(traj) + 1 and len(actions) == len(rewards) == len(traj)
    e2.reference = list(traj[0])
    obs = e2.vector_reset_tensor().clone()
    np.testing.assert_allclose(observations[0], obs[0].cpu().numpy(), atol=1e-6)
    prev, tr = None, None
    for k, x in enumerate(traj):
        a = pol.forward(obs, prev, tr)
        e2.reference = list(x)
        ob, rw, trn = e2.vector_step_tensor(a)
        obs, prev, tr = ob.clone(), a, trn.clone()
        np.testing.assert_allclose(observations[k + 1], obs[0].cpu().numpy(), atol=2e-5, err_msg="k=%d" % k)
        np.testing.assert_allclose(actions[k], a[0].cpu().numpy(), atol=2e-5)
        assert abs(rewards[k] - float(rw[0])) < 2e-5
    assert list(e1.reference) == [float(v) for v in traj[-1]]


def test_custom_lstm_recurrent_actor_vs_reference_model(PG):
    """CustomLSTM (the LSTM in the action path): stepped one observation at a time on the device against the reference
    model's outputs over 24-step episodes"""
    from mujoco_drone_amd.policy import DevicePolicy
    w = weights_of(PG, "custom_lstm")
    pol = DevicePolicy("CustomLSTM", w)
    assert pol.has_history
    o, a = PG["custom_lstm_obs_seq"], PG["custom_lstm_action_seq"]
    pol.reset_state(o.shape[0])
    for t in range(o.shape[1]):
        prev = torch.tensor(a[:, t - 1], device="cuda") if t > 0 else None
        _, logits, value = pol.forward(torch.tensor(o[:, t], device="cuda"), prev, None, counter=t, want_logits=True, want_value=True)
        np.testing.assert_allclose(logits.cpu().numpy(), PG["custom_lstm_logits"][:, t], atol=3e-5, err_msg="t=%d" % t)
        np.testing.assert_allclose(value.cpu().numpy(), PG["custom_lstm_value"][:, t], atol=3e-5)


@pytest.mark.parametrize("tag,family", [("rma_smaller", "RMA_model_smaller"), ("rma_smaller2", "RMA_model_smaller2")])
def test_smaller_rma_variants_vs_reference_models(PG, tag, family, kernel):
    """RMA_model_smaller / RMA_model_smaller2 (residual blocks in the value head folded into the following layer's weights)"""
    from mujoco_drone_amd.policy import DevicePolicy
    pol = DevicePolicy(family, weights_of(PG, tag))
    assert (pol.kernel > 0) == (kernel == "specialised")
    obs, prev = torch.tensor(PG["obs"], device="cuda"), torch.tensor(PG["prev_actions"], device="cuda")
    act, logits, value = pol.forward(obs, prev, want_logits=True, want_value=True)
    np.testing.assert_allclose(logits.cpu().numpy(), PG[tag + "_logits"], atol=2e-5)
    np.testing.assert_allclose(value.cpu().numpy(), PG[tag + "_value"], atol=2e-5)
    np.testing.assert_allclose(pol.embedding(obs, prev).cpu().numpy(), PG[tag + "_z"], atol=1e-5)
    np.testing.assert_array_equal(pol.forward(obs, prev).cpu().numpy(), act.cpu().numpy())


@pytest.mark.parametrize("tag,family", [("lstm_bigger", "CustomLSTMbigger"), ("lstm_common_f", "CustomLSTMbiggerCommonF"),
                                        ("dsn_lstm", "DSN_LSTM_model")])
def test_recurrent_variants_vs_reference_models(PG, tag, family, kernel):
    """CustomLSTMbigger / CustomLSTMbiggerCommonF / DSN_LSTM_model (its three per-axis LSTMs run as one block-diagonal LSTM):
    stepped one observation at a time against the reference models' outputs over 24-step episodes; then a second episode on
    half of the envs (prev_truncated) restarts from the zero state"""
    from mujoco_drone_amd.policy import DevicePolicy
    pol = DevicePolicy(family, weights_of(PG, tag))
    assert pol.has_history and (pol.kernel > 0) == (kernel == "specialised")
    o, a = PG[tag + "_obs_seq"], PG[tag + "_action_seq"]
    n, T = o.shape[0], o.shape[1]
    pol.reset_state(n)
    for t in range(T):
        prev = torch.tensor(a[:, t - 1], device="cuda") if t > 0 else None
        _, logits, value = pol.forward(torch.tensor(o[:, t], device="cuda"), prev, None, counter=t, want_logits=True, want_value=True)
        np.testing.assert_allclose(logits.cpu().numpy(), PG[tag + "_logits"][:, t], atol=3e-5, err_msg="t=%d" % t)
        np.testing.assert_allclose(value.cpu().numpy(), PG[tag + "_value"][:, t], atol=3e-5)
    fresh = np.zeros(n, dtype=np.uint8)
    fresh[::2] = 1
    _, logits, _ = pol.forward(torch.tensor(o[:, 0], device="cuda"), torch.tensor(a[:, T - 1], device="cuda"),
                               torch.tensor(fresh, device="cuda"), counter=T, want_logits=True, want_value=True)
    got = logits.cpu().numpy()
    np.testing.assert_allclose(got[::2], PG[tag + "_logits"][::2, 0], atol=3e-5)
    assert np.abs(got[1::2] - PG[tag + "_logits"][1::2, 0]).max() > 1e-3        # the others carry their state on


def _logits_passthrough(p, D, ns, npar, na):
    """a one-op layer program: the observation row IS the logits row (to drive the output stage with chosen logits)"""
    p.copy_obs(0, 2 * na, 0, 0)
    p._put(np.zeros(4, dtype=np.float32))
    return dict(widths=[16], logits=(0, 0, 2 * na))


def test_squashed_gaussian_output_stage_vs_reference(PG):
    """MySquashedGaussian on chosen logits: deterministic action and its log-probability against the reference class's own
    numbers; sampled actions are sigmoid(N(mean, std)) (KS test on the standardised pre-squash values), reproducible from the
    seed, and their log-probability equals the oracle's on the same (logits, action)"""
    from scipy import stats
    from mujoco_drone_amd.policy import DevicePolicy
    from oracle import policy_ref as P
    pol = DevicePolicy(_logits_passthrough, {}, obs_dim=8, dist="MySquashedGaussian")
    lg = torch.tensor(PG["sg_logits"], device="cuda")
    act, logp = pol.forward(lg, want_logp=True)
    np.testing.assert_allclose(act.cpu().numpy(), PG["sg_action"], atol=2e-6)
    np.testing.assert_allclose(logp.cpu().numpy(), PG["sg_logp_action"], rtol=2e-5, atol=2e-4)
    n = 20000
    rng = np.random.default_rng(4)
    big = np.concatenate([rng.normal(scale=1.0, size=(n, 4)), rng.uniform(-1.5, 0.5, (n, 4))], axis=1).astype(np.float32)
    bl = torch.tensor(big, device="cuda")
    x1, lp1 = pol.forward(bl, explore=True, seed=7, counter=3, want_logp=True)
    x2, _ = pol.forward(bl, explore=True, seed=7, counter=3, want_logp=True)
    x3 = pol.forward(bl, explore=True, seed=7, counter=4)
    assert torch.equal(x1, x2) and not torch.equal(x1, x3)
    x = x1.cpu().numpy().astype(np.float64)
    assert x.min() > 0 and x.max() < 1
    z = (np.log(x / (1 - x)) - big[:, :4]) / np.exp(big[:, 4:])
    for c in range(4):
        assert stats.kstest(z[:, c], "norm").pvalue > 1e-3, c
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.03
    np.testing.assert_allclose(lp1.cpu().numpy(), P.squashed_gaussian_logp(big, x1.cpu().numpy()), rtol=2e-4, atol=2e-3)
    with pytest.raises(ValueError):
        DevicePolicy(_logits_passthrough, {}, obs_dim=8, dist="Gaussian")


def test_evaluate_trajectory_lstmest_history_window_like_reference(PG):
    """evaluation.py:76-132 rolls a 32-step observation / action history by hand (zeros before the episode start, newest row
    last, the action row = the action taken BEFORE that observation) and feeds it to the estimator network each step.  The device
    keeps that history in per-env rings; its actions along a 45-step waypoint trajectory must equal the float64 oracle network fed
    with the hand-rolled windows built from the device's own observations and actions."""
    from mujoco_drone_amd.policy import DevicePolicy
    from mujoco_drone_amd.evaluation import evaluate_trajectory_lstmest, gen_ramp_trajectory
    from mujoco_drone_amd.environments.BaseDroneEnv import base_config
    from mujoco_drone_amd.environments.observation_wrappers import LocalFrameFullStateEnv
    from mujoco_drone_amd.environments.rewards import distance_energy_reward_pendulum_en4
    from oracle import policy_ref as P
    w = weights_of(PG, "cnn_est_hist")
    pol = DevicePolicy("CNNestimator_estimate", w, obs_dim=23, num_states=23)
    n, Lw = 6, 32
    cfg = dict(base_config, num_drones=n, reward_fcn=distance_energy_reward_pendulum_en4, random_params=False, state_difficulty=0.3,
               max_steps=4096, max_distance=1e9)
    env = LocalFrameFullStateEnv(cfg)
    _, traj = gen_ramp_trajectory(0.1, 0.45, [0, 0, 15, 0], [0.5, -0.3, 15.4, 0.2])
    observations, actions, rewards, out = evaluate_trajectory_lstmest(env, pol, traj)
    T = len(traj)
    assert T == 45 and len(observations) == T + 1
    dev_obs = out["obs"].cpu().numpy().astype(np.float64)          # [T, n, 23]: observation AFTER step t
    dev_act = out["actions"].cpu().numpy().astype(np.float64)      # [T, n, 4]:  action of step t
    # drone 0 (the one the reference's function returns): the observation it acted on at step t is the reset observation for
    # t = 0 and dev_obs[t - 1] afterwards; the action row next to it is the action taken before it (zeros at the start)
    hist_o, hist_a = np.zeros((1, Lw, 23)), np.zeros((1, Lw, 4))
    for t in range(T):
        hist_o[:, :-1] = hist_o[:, 1:].copy(); hist_a[:, :-1] = hist_a[:, 1:].copy()
        hist_o[0, -1] = np.asarray(observations[0], dtype=np.float64) if t == 0 else dev_obs[t - 1, 0]
        hist_a[0, -1] = 0.0 if t == 0 else dev_act[t - 1, 0]
        logits, _, _ = P.cnn_estimator_hist(w, hist_o, hist_a)
        np.testing.assert_allclose(dev_act[t, 0], P.beta_mean_action(logits)[0], atol=3e-5, err_msg="t=%d" % t)
    np.testing.assert_allclose(np.asarray(actions), dev_act[:, 0], atol=0)


def test_collection_loop_example_runs():
    """examples/collect_fragments.py: env + exploring actor + batch / episode statistics, fragment after fragment, on one GPU"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import importlib.util
    spec = importlib.util.spec_from_file_location("collect_fragments", os.path.join(os.path.dirname(HERE), "examples", "collect_fragments.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    log = mod.main(num_envs=512, fragment=64, fragments=3, quiet=True)
    assert len(log) == 3 and all(np.isfinite(x[2]) and x[3] > 0 for x in log)
