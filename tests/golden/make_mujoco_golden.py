#!/usr/bin/env python3
"""One-command pin of the physics step against the reference's own MuJoCo path (build container only).

    python tests/golden/make_mujoco_golden.py [--reference /root/reference] [--out tests/golden/mujoco_trajectories.npz]

Runs ONLY where `import mujoco, dm_control, gymnasium` succeeds (none of them is in the round-1/2 images: then the script
says so and exits 2 without writing anything).  It imports the reference's own classes from --reference (nothing of the
reference is copied: only numbers are stored), steps them for the BASELINE configurations and stores, in float64 and per step,
what `mujoco.mj_step` left in `data.qpos / qvel / act / sensordata` (environments/mujoco_vecenv.py:404-413) together with every
input needed to replay the run elsewhere (per-drone parameters, initial state, actions) and the env's own outputs:

  config1   SimpleDrone(num_drones=1), init qpos0, action 0.7 x 4, 200 steps                      (test_env.py:7-14)
  config2   SimpleDrone(num_drones=16), init qpos0 (spawn grid at z = 0.15 m, just above the floor), actions U[0.5,1),
            200 steps                                                                               (SimpleDrone.py:41-46)
  config3   LocalFrameRPYParamsEnv, 64 drones + loads, random parameters (difficulty 1), state_difficulty 0.2,
            distance_energy_reward, actions U[0,1), 200 steps, no resets                            (train_RMA.py:66-75)
  config5   LocalFrameFullStateEnv, 64 drones + loads, default parameters, state_difficulty 0.8,
            distance_energy_reward_pendulum_en4, static reference, 200 steps                        (train_LSTM.py:37,41,70,78)
  floor     BaseDroneEnv, 4 drones + loads released at z = 1.6 m with zero rotor commands, 300 steps: the load, then the
            tether and the airframe land on the floor plane                                                                  (env_gen.py:97)

tests/test_mujoco_pin.py compares the oracle (CPU) and the HIP kernels (GPU) with this file when it exists and skips, printing
why, when it does not.  ray is only needed for the `VectorEnv` base class name (BaseDroneEnv.py:53): if it is absent a
placeholder class stands in for it; gymnasium, mujoco and dm_control are used for real.
"""
import argparse
import os
import sys
import types

import numpy as np


def need(mod):
    try:
        return __import__(mod)
    except Exception as ex:  # ordinary import errors: the packages are simply not installed
        print("make_mujoco_golden: cannot import %s (%s: %s) -- nothing written; run this where mujoco, dm_control and "
              "gymnasium are installed" % (mod, type(ex).__name__, ex))
        sys.exit(2)


class EnvContext(dict):
    """what RLlib hands to the env: a dict with a worker_index attribute (> train_vis, so no viewer window is opened:
    BaseDroneEnv.py:62-66)"""

    def __init__(self, cfg, worker_index=1):
        dict.__init__(self, cfg)
        self.worker_index = worker_index


def snapshot(env):
    d = env.data
    return (np.array(d.qpos, dtype=np.float64), np.array(d.qvel, dtype=np.float64), np.array(d.act, dtype=np.float64),
            np.array(d.sensordata, dtype=np.float64))


def run_vector_env(env, actions, out, key):
    """env: a BaseDroneEnv (sub)class instance already reset; actions [T, N, 4]"""
    n = env.num_drones
    raw = np.array([[p['mass'], p['arm_len'], p['motor_force'], p['motor_tau'], p['pendulum_len'], p['weight_mass']]
                    for p in env.drone_params], dtype=np.float64)
    q0, v0, a0, s0 = snapshot(env)
    rec = {k: [] for k in ("qpos", "qvel", "act", "sensordata", "obs", "reward", "truncated")}
    for t in range(actions.shape[0]):
        obs, rew, term, trunc, info = env.vector_step([actions[t, i] for i in range(n)])
        q, v, a, s = snapshot(env)
        rec["qpos"].append(q); rec["qvel"].append(v); rec["act"].append(a); rec["sensordata"].append(s)
        rec["obs"].append(np.array(obs, dtype=np.float64)); rec["reward"].append(np.array(rew, dtype=np.float64))
        rec["truncated"].append(np.array(trunc, dtype=bool))
    out[key + "_raw"] = raw
    out[key + "_qpos0"], out[key + "_qvel0"], out[key + "_act0"], out[key + "_sensordata0"] = q0, v0, a0, s0
    out[key + "_actions"] = actions
    out[key + "_reference"] = np.array(env.reference, dtype=np.float64)
    out[key + "_timestep"] = np.float64(env.model.opt.timestep)
    out[key + "_frame_skip"] = np.int64(env.frame_skip)
    for k, v in rec.items():
        out[key + "_" + k] = np.array(v)


def run_simple(SimpleDrone, n, actions, out, key):
    env = SimpleDrone(num_drones=n)
    env.set_state(env.init_qpos.copy(), env.init_qvel.copy())      # exactly qpos0 (reset_model would add noise), act = 0
    q0, v0, a0, s0 = snapshot(env)
    rec = {k: [] for k in ("qpos", "qvel", "act", "sensordata", "obs", "reward", "terminated")}
    for t in range(actions.shape[0]):
        ob, rew, term, info = env.step(actions[t])
        q, v, a, s = snapshot(env)
        rec["qpos"].append(q); rec["qvel"].append(v); rec["act"].append(a); rec["sensordata"].append(s)
        rec["obs"].append(np.array(ob, dtype=np.float64)); rec["reward"].append(np.float64(rew)); rec["terminated"].append(bool(term))
    out[key + "_raw"] = np.tile(np.array([1.35, 0.15, 7.5, 0.015, 0.0, 0.0]), (n, 1))     # env_gen.py:26-32 defaults
    out[key + "_qpos0"], out[key + "_qvel0"], out[key + "_act0"], out[key + "_sensordata0"] = q0, v0, a0, s0
    out[key + "_actions"] = actions
    out[key + "_reference"] = np.array(list(env.reference) + [0.0], dtype=np.float64)[:4]
    out[key + "_timestep"] = np.float64(env.model.opt.timestep)
    out[key + "_frame_skip"] = np.int64(env.frame_skip)
    for k, v in rec.items():
        out[key + "_" + k] = np.array(v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "mujoco_trajectories.npz"))
    args = ap.parse_args()
    mujoco = need("mujoco")
    need("dm_control")
    need("gymnasium")
    try:
        import ray.rllib.env.vector_env  # noqa: F401
    except Exception:
        pkg = [types.ModuleType(n) for n in ("ray", "ray.rllib", "ray.rllib.env", "ray.rllib.env.vector_env")]
        for a, b in zip(pkg, pkg[1:]):
            setattr(a, b.__name__.split(".")[-1], b)
        pkg[-1].VectorEnv = type("VectorEnv", (), {"__init__": lambda self, observation_space, action_space, num_envs: None})
        for m in pkg:
            sys.modules[m.__name__] = m
    if not os.path.isdir(args.reference):
        print("make_mujoco_golden: reference tree %s not found" % args.reference)
        sys.exit(2)
    sys.path.insert(0, args.reference)
    from environments.SimpleDrone import SimpleDrone
    from environments.BaseDroneEnv import BaseDroneEnv, base_config
    from environments import observation_wrappers as ow
    from environments import rewards

    out = {"mujoco_version": np.array(mujoco.__version__)}
    # SimpleDrone as written asserts round(1/dt) == render_fps with dt = 2 ms and render_fps = 50 (mujoco_env_custom.py:122-124):
    # the class attribute is set to the value the assertion wants; nothing else changes
    SimpleDrone.metadata = dict(SimpleDrone.metadata, render_fps=500)
    rng = np.random.default_rng(2024)
    run_simple(SimpleDrone, 1, np.full((200, 4), 0.7), out, "config1")
    run_simple(SimpleDrone, 16, rng.uniform(0.5, 1.0, (200, 64)), out, "config2")

    cfg = dict(base_config)
    cfg.update(num_drones=64, random_params=True, param_difficulty=1, state_difficulty=0.2, max_steps=1024,
               regen_env_at_steps=None, reward_fcn=rewards.distance_energy_reward, train_vis=0)
    env = ow.LocalFrameRPYParamsEnv(EnvContext(cfg))
    env.vector_reset()
    run_vector_env(env, rng.uniform(0.0, 1.0, (200, 64, 4)), out, "config3")

    cfg = dict(base_config)
    cfg.update(num_drones=64, random_params=False, state_difficulty=0.8, max_steps=1024, regen_env_at_steps=None,
               reward_fcn=rewards.distance_energy_reward_pendulum_en4, train_vis=0)
    env = ow.LocalFrameFullStateEnv(EnvContext(cfg))
    env.vector_reset()
    run_vector_env(env, rng.uniform(0.0, 1.0, (200, 64, 4)), out, "config5")

    cfg = dict(base_config)
    cfg.update(num_drones=4, random_params=False, max_steps=10 ** 6, regen_env_at_steps=None, max_distance=1e9, train_vis=0)
    env = BaseDroneEnv(EnvContext(cfg))
    env.vector_reset()
    qpos, qvel = env.data.qpos.copy(), env.data.qvel.copy()
    nq = qpos.size // 4
    for i in range(4):
        qpos[nq * i + 2] = 1.6 + 0.05 * i                     # release heights: the load (1.2 m tether) hangs ~0.3 m above the floor
        qpos[nq * i + 3:nq * i + 7] = [1, 0, 0, 0]
        qpos[nq * i + 7:nq * i + 9] = [0.3 * (i - 1.5), 0.2]
    qvel[:] = 0
    env.set_state(qpos, qvel)
    run_vector_env(env, np.zeros((300, 4, 4)), out, "floor")

    np.savez_compressed(args.out, **out)
    print("wrote %s (%d arrays, MuJoCo %s)" % (args.out, len(out), mujoco.__version__))


if __name__ == "__main__":
    main()
