"""evaluation.py's trajectory evaluation on the GPU: the waypoint generators (evaluation.py:135-152) and
evaluate_trajectory (:38-72) -- reset at the first waypoint, then for every waypoint x: action = policy(obs),
env.reference = x, vector_step -- as ONE device-side rollout (qd_set_reference_schedule + qd_rollout_policy)."""
import ctypes as C

import numpy as np

from . import _lib as L


def gen_circle_trajectory(T=10, f=0.5, r=1, h=1, dt=0.01):
    """evaluation.py:135-138 (waypoint arrays are host-side configuration, not part of the compute path)"""
    t = np.arange(0, T, dt)
    return t, np.stack([r * np.cos(2 * np.pi * f * t), r * np.sin(2 * np.pi * f * t), h * np.ones_like(t), np.zeros_like(t)], axis=1)


def gen_step_trajectory(step_time=5, duration=10, start_pos=(0, 0, 0, 0), end_pos=(0, 0, 1, 0), dt=0.01):
    """evaluation.py:141-144"""
    t = np.arange(0, duration, dt)
    return t, np.where((t < step_time)[:, None], np.asarray(start_pos, float), np.asarray(end_pos, float))


def gen_ramp_trajectory(start_time=5, duration=10, start_pos=(0, 0, 0, 0), end_pos=(0, 0, 1, 0), dt=0.01):
    """evaluation.py:147-152"""
    t = np.arange(0, duration, dt)
    s, e = np.asarray(start_pos, float), np.asarray(end_pos, float)
    w = np.where(t < start_time, 0.0, (t - start_time) / (duration - start_time))
    return t, s + w[:, None] * (e - s)


def evaluate_trajectory(env, policy, trajectory, explore=False, seed=0):
    """Returns (observations, actions, rewards) of drone 0 like the reference (observations has one more entry: the reset
    observation), plus the full device-side fragment dict (all drones) as a fourth value."""
    traj = np.ascontiguousarray(np.asarray(trajectory, dtype=np.float64).reshape(-1, 4))
    env.reference = [float(x) for x in traj[0]]                              # :44-47
    obs0 = env.vector_reset_tensor().clone()
    policy.reset_state(env.num_drones)
    lib = L.lib()
    L.check(lib.qd_set_reference_schedule(env._dev.handle, traj.ctypes.data_as(C.POINTER(C.c_double)), len(traj)))
    try:
        out = policy.rollout(env._dev, len(traj), obs0, explore=explore, seed=seed)
    finally:
        L.check(lib.qd_set_reference_schedule(env._dev.handle, None, 0))
    env._reference = [float(x) for x in traj[-1]]                            # what the scheduled rollout left on the device
    env._ref_pushed = env._reference
    env.total_steps += len(traj)
    o = out["obs"][:, 0].cpu().numpy().astype(np.float64)
    observations = [obs0[0].cpu().numpy().astype(np.float64)] + list(o)
    actions = list(out["actions"][:, 0].cpu().numpy().astype(np.float64))
    rewards = [float(x) for x in out["reward"][:, 0].cpu().numpy()]
    return observations, actions, rewards, out


def evaluate_trajectory_lstmest(env, policy, trajectory, explore=False, seed=0):
    """evaluation.py:76-132 rolls a (seq_len)-step observation / action history by hand on the host for the estimator networks; a
    windowed `DevicePolicy` (`has_history`) keeps that history in per-env rings on the device, so this is `evaluate_trajectory` --
    including the waypoint switching the reference's version left commented out (:106-110)."""
    return evaluate_trajectory(env, policy, trajectory, explore=explore, seed=seed)


def load_policy_state(checkpoint):
    """evaluation.py:155-159: RLlib's pickled policy state of a checkpoint directory; its 'weights' dict is what `DevicePolicy` takes"""
    import os
    import pickle
    with open(os.path.join(checkpoint, 'policies/default_policy/policy_state.pkl'), 'rb') as f:
        return pickle.load(f)
