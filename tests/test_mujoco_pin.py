"""Physics parity against the reference's own MuJoCo path -- DORMANT until tests/golden/mujoco_trajectories.npz exists.

The file is written by tests/golden/make_mujoco_golden.py on a machine where `mujoco`, `dm_control` and `gymnasium` import
(they do not in this image; SURVEY 8c: ordinary import errors).  Until then every test here skips and says why; the oracle's
restatement of mj_step stays "parity unpinned" (DESIGN.md section 2).  With the file present:
  CPU   the float64 oracle replays each recorded run (same parameters, initial state, actions) and must stay within the
        BASELINE bar of MuJoCo's qpos / qvel / act over the 200 recorded steps; sensordata and the env outputs are compared too
  GPU   the HIP kernels do the same through the C ABI
Reports are per component group: pure absolute, pure relative (to the group's scale) and the element-wise mixed measure.
"""
import os

import numpy as np
import pytest

from divergence import Divergence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "mujoco_trajectories.npz")
WHY = ("tests/golden/mujoco_trajectories.npz is absent: generate it with `python tests/golden/make_mujoco_golden.py` where "
       "mujoco + dm_control + gymnasium are installed (physics parity stays UNPINNED until then)")
BAR = 1e-4          # BASELINE.json: <= 1e-4 relative state divergence vs MuJoCo over 200 steps
# The floor drop goes through impacts: a contact that closes one step earlier or later changes the angular rates by 1e-4 rad/s
# at rates of 9 rad/s.  For that case the bar is 1e-4 on the pure relative measure and 1e-3 on the element-wise mixed one.
BAR_MIXED = {"floor": 1e-3}
CASES = {  # key: (load, obs variant, reward, ctrl map (1 = 0.1 + 0.9 a), floor contact)
    "config1": (False, "SimpleDrone", "simple_drone_reward", 0, False),
    "config2": (False, "SimpleDrone", "simple_drone_reward", 0, True),
    "config3": (True, "LocalFrameRPYParamsEnv", "distance_energy_reward", 1, False),
    "config5": (True, "LocalFrameFullStateEnv", "distance_energy_reward_pendulum_en4", 1, False),
    "floor": (True, "BaseDroneEnv", "default_reward_fcn", 1, True),
}


@pytest.fixture(scope="module")
def traj():
    if not os.path.exists(GOLDEN):
        print("\nSKIP: " + WHY)
        pytest.skip(WHY)
    return np.load(GOLDEN, allow_pickle=False)


def case(traj, key):
    load = CASES[key][0]
    nq, nv = (9, 8) if load else (7, 6)
    g = lambda name: traj[key + "_" + name]
    n = g("raw").shape[0]
    T = g("actions").shape[0]
    return dict(load=load, n=n, T=T, raw=g("raw"), h=float(g("timestep")), frame_skip=int(g("frame_skip")), ref=g("reference"),
                qpos0=g("qpos0").reshape(n, nq), qvel0=g("qvel0").reshape(n, nv), act0=g("act0").reshape(n, 4),
                actions=g("actions").reshape(T, n, 4), qpos=g("qpos").reshape(T, n, nq), qvel=g("qvel").reshape(T, n, nv),
                act=g("act").reshape(T, n, 4), sens=g("sensordata").reshape(T, n, 3))


def test_fixture_is_self_consistent(traj):
    assert str(traj["mujoco_version"])
    for key in CASES:
        c = case(traj, key)
        assert c["qpos"].shape[0] == c["T"] >= 200 and np.isfinite(c["qpos"]).all() and np.isfinite(c["qvel"]).all()
        assert np.allclose(np.linalg.norm(c["qpos"][:, :, 3:7], axis=-1), 1.0, atol=1e-9)      # MuJoCo renormalises every step
        assert np.all(c["act0"] == 0)                                                           # fresh MjData


def oracle_replay(orc, c, ctrl_map, floor):
    """step the oracle through a recorded run; yields (t, qpos, qvel, act, sensordata) after every step"""
    models = [orc.build_model(r) for r in c["raw"]]
    q, v, a = c["qpos0"].copy(), c["qvel0"].copy(), c["act0"].copy()
    s = np.zeros((c["n"], 3))
    stepper = orc.step_floor if floor else orc.step
    for t in range(c["T"]):
        ctrl = np.clip(0.1 + 0.9 * c["actions"][t] if ctrl_map else c["actions"][t], 0.0, 1.0)
        for i in range(c["n"]):
            q[i], v[i], a[i], s[i] = stepper(models[i], c["h"], c["frame_skip"], q[i], v[i], a[i], ctrl[i])[:4]
        yield t, q, v, a, s


def check_oracle(traj, orc, key):
    load, obs, rew, ctrl_map, floor = CASES[key]
    c = case(traj, key)
    div, sens_err = Divergence(load), 0.0
    for t, q, v, a, s in oracle_replay(orc, c, ctrl_map, floor):
        sens_err = max(sens_err, float(np.abs(s - c["sens"][t]).max()))
        if t < 200:
            div.update(dict(qpos=q, qvel=v, act=a), dict(qpos=c["qpos"][t], qvel=c["qvel"][t], act=c["act"][t]))
    print("\n" + div.table("%s: oracle vs MuJoCo %s, %d envs, first 200 of %d steps" % (key, traj["mujoco_version"], c["n"], c["T"])))
    print("accelerometer: max |delta| = %.2e" % sens_err)
    assert div.max("mixed") <= BAR_MIXED.get(key, BAR) and div.max("rel") <= BAR, "oracle restatement of mj_step diverges from MuJoCo"


@pytest.mark.parametrize("key", list(CASES))
def test_oracle_vs_mujoco(traj, orc, key):
    check_oracle(traj, orc, key)


@pytest.mark.gpu
@pytest.mark.parametrize("key", list(CASES))
def test_hip_vs_mujoco(traj, key):
    check_hip(traj, key)


def check_hip(traj, key):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mujoco_drone_amd import _lib as L
    from mujoco_drone_amd.environments import _device
    load, obs, rew, ctrl_map, floor = CASES[key]
    c = case(traj, key)
    cfg = L.QdConfig()
    cfg.num_envs, cfg.model = c["n"], (L.MODEL_LOAD if load else L.MODEL_NOLOAD)
    cfg.obs_kind, cfg.reward_kind = L.OBS_KINDS.index(obs), L.REWARD_KINDS.index(rew)
    cfg.frame_skip, cfg.max_steps, cfg.ctrl_map = c["frame_skip"], 10 ** 6, ctrl_map
    cfg.term_kind = L.TERM_SIMPLE if obs == "SimpleDrone" else L.TERM_DEFAULT
    cfg.random_start, cfg.random_params, cfg.auto_reset, cfg.floor_contact = L.START_FIXED, 0, 0, int(floor)
    cfg.timestep, cfg.max_distance = c["h"], 1e9
    for k in range(4):
        cfg.reference[k] = float(c["ref"][k]); cfg.start_pos[k] = float(c["ref"][k])
    cfg.seed = 42
    env = _device.DeviceEnv(cfg)
    env.set_params(c["raw"])
    env.set_state(c["qpos0"], c["qvel0"], c["act0"])
    div, sens_err = Divergence(load), 0.0
    for t in range(c["T"]):
        env.step(torch.as_tensor(c["actions"][t], dtype=torch.float32, device="cuda"))
        if t < 200 and (t % 10 == 9 or t == 199):
            q, v, a, s, _ = [x.cpu().numpy().astype(np.float64) for x in env.get_state()]
            div.update(dict(qpos=q, qvel=v, act=a), dict(qpos=c["qpos"][t], qvel=c["qvel"][t], act=c["act"][t]))
            sens_err = max(sens_err, float(np.abs(s - c["sens"][t]).max()))
    print("\n" + div.table("%s: HIP vs MuJoCo %s, %d envs, first 200 of %d steps" % (key, traj["mujoco_version"], c["n"], c["T"])))
    print("accelerometer: max |delta| = %.2e" % sens_err)
    assert div.max("mixed") <= BAR_MIXED.get(key, BAR) and div.max("rel") <= BAR, "HIP step diverges from MuJoCo"


# ---- the harness itself, exercised without MuJoCo -------------------------------------------------------------------
# A stand-in fixture with the same keys and shapes as make_mujoco_golden.py writes, filled by the ORACLE (so it pins nothing
# about MuJoCo): it keeps the dormant tests above from rotting -- array layouts, the replay of parameters / initial state /
# actions, the floor cases -- and on the GPU it is one more HIP-vs-oracle trajectory check, floor contact included.
def synthetic_fixture(orc, path, n_load=3, n_simple=2, T=200):
    rng = np.random.default_rng(5)
    out = {"mujoco_version": np.array("none (oracle stand-in)")}

    def put(key, raw, qpos0, qvel0, actions, h, frame_skip, ref, T_):
        load, obs, rew, ctrl_map, floor = CASES[key]
        n = raw.shape[0]
        c = dict(load=load, n=n, T=T_, raw=raw, h=h, frame_skip=frame_skip, qpos0=qpos0, qvel0=qvel0, act0=np.zeros((n, 4)),
                 actions=actions)
        rec = {k: [] for k in ("qpos", "qvel", "act", "sensordata")}
        for t, q, v, a, s in oracle_replay(orc, c, ctrl_map, floor):
            rec["qpos"].append(q.ravel().copy()); rec["qvel"].append(v.ravel().copy()); rec["act"].append(a.ravel().copy())
            rec["sensordata"].append(s.ravel().copy())
        out.update({key + "_raw": raw, key + "_qpos0": qpos0.ravel(), key + "_qvel0": qvel0.ravel(), key + "_act0": np.zeros(4 * n),
                    key + "_actions": actions.reshape(T_, -1) if not load else actions, key + "_reference": np.asarray(ref, dtype=np.float64),
                    key + "_timestep": np.float64(h), key + "_frame_skip": np.int64(frame_skip)})
        for k, v in rec.items():
            out[key + "_" + k] = np.array(v)

    simple = lambda n: np.tile(np.array([1.35, 0.15, 7.5, 0.015, 0.0, 0.0]), (n, 1))
    q1 = np.array([[0, 0, 1.0, 1, 0, 0, 0]]); put("config1", simple(1), q1, np.zeros((1, 6)), np.full((T, 1, 4), 0.7), 0.001, 2, [0, 0, 1, 0], T)
    q2 = np.array([[0.25 * i, 0.0, 0.15, 1, 0, 0, 0] for i in range(n_simple)])
    put("config2", simple(n_simple), q2, np.zeros((n_simple, 6)), rng.uniform(0.5, 1.0, (T, n_simple, 4)), 0.001, 2, [0, 0, 1, 0], T)
    center, width = np.array([1, 0.17, 7, 0.01, 1.2, 0.3]), np.array([0.1, 0.02, 1, 0.0025, 0.2, 0.05])
    for key, raw in (("config3", center + rng.uniform(-1, 1, (n_load, 6)) * width), ("config5", np.tile(center, (n_load, 1)))):
        q = np.zeros((n_load, 9)); q[:, :3] = [0, 0, 15] + rng.uniform(-0.3, 0.3, (n_load, 3)); q[:, 3] = 1; q[:, 7:] = rng.normal(0, 0.1, (n_load, 2))
        put(key, raw, q, rng.normal(0, 0.2, (n_load, 8)), rng.uniform(0.3, 0.7, (T, n_load, 4)), 0.01, 1, [0, 0, 15, 0], T)
    qf = np.zeros((2, 9)); qf[:, 2] = [1.6, 1.65]; qf[:, 3] = 1; qf[:, 7:] = [[-0.3, 0.2], [0.3, 0.2]]
    put("floor", np.tile(center, (2, 1)), qf, np.zeros((2, 8)), np.zeros((T + 100, 2, 4)), 0.01, 1, [0, 0, 15, 0], T + 100)
    np.savez(path, **out)
    return np.load(path, allow_pickle=False)


def test_pin_harness_runs_on_an_oracle_made_fixture(orc, tmp_path):
    fx = synthetic_fixture(orc, str(tmp_path / "standin.npz"), n_load=2, n_simple=2, T=200)
    test_fixture_is_self_consistent(fx)
    for key in CASES:
        check_oracle(fx, orc, key)          # the oracle against its own recording: exact, what is tested is the plumbing


@pytest.mark.gpu
def test_hip_vs_oracle_through_the_pin_harness(orc, tmp_path):
    fx = synthetic_fixture(orc, str(tmp_path / "standin.npz"), n_load=4, n_simple=3, T=200)
    for key in CASES:
        check_hip(fx, key)
