"""diagnostic (not a test): soak of the floor contact -- 4096 envs per model thrown at the floor with random attitudes, spins and
rotor commands, thousands of steps: nothing may become non-finite, fall through the floor or gain energy without bound"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mujoco_drone_amd import _lib as L
from mujoco_drone_amd.environments import _device as dev
from test_gpu_parity import make_cfg, rand_raw
n, steps = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(1)
for load in (False, True):
    c = make_cfg(L, n, load=load, obs="BaseDroneEnv", reward="default_reward_fcn", frame_skip=1, h=0.002, ctrl_map=0, max_steps=10 ** 7, max_distance=1e9)
    c.floor_contact = 1
    env = dev.DeviceEnv(c)
    env.set_params(rand_raw(rng, n, load))
    nq, nv = (9, 8) if load else (7, 6)
    qpos = np.zeros((n, nq)); q = rng.normal(size=(n, 4)); qpos[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    qpos[:, :2] = rng.normal(scale=1.0, size=(n, 2)); qpos[:, 2] = rng.uniform(0.1, 2.0, n)
    if load:
        qpos[:, 7:] = rng.normal(scale=0.6, size=(n, 2))
    qvel = rng.normal(scale=2.0, size=(n, nv)); qvel[:, 2] -= 1.0
    env.set_state(qpos, qvel, np.zeros((n, 4)))
    g = torch.Generator(device="cuda").manual_seed(2)
    worst_low, vmax = 0.0, 0.0
    for t in range(steps):
        if t % 50 == 0:
            a = torch.rand((n, 4), generator=g, device="cuda") * (0.6 if t < steps // 2 else 0.0)   # second half: rotors off, everything settles
        env.step(a)
        if t % 250 == 249 or t == steps - 1:
            q, v = env.get_state()[:2]
            assert bool(torch.isfinite(q).all()) and bool(torch.isfinite(v).all()), t
            worst_low = min(worst_low, float(q[:, 2].min())); vmax = max(vmax, float(v.abs().max()))
    q, v = env.get_state()[:2]
    print("%-8s: %d envs x %d steps ok; lowest origin height ever %.4f m, max |qvel| seen %.1f, at the end: max |qvel| %.3f, origin heights %.3f..%.3f"
          % ("load" if load else "no load", n, steps, worst_low, vmax, float(v.abs().max()), float(q[:, 2].min()), float(q[:, 2].max())), flush=True)
