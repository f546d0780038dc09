"""Diagnostic (not a test): throughput of the reference-faithful list API (RLlib VectorEnv layout)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_drone_amd.environments.BaseDroneEnv import base_config
from mujoco_drone_amd.environments.observation_wrappers import LocalFrameRPYParamsEnv
from mujoco_drone_amd.environments.rewards import distance_energy_reward
for n in (64, 4096):
    cfg = dict(base_config, num_drones=n, reward_fcn=distance_energy_reward, param_difficulty=1, state_difficulty=0.2,
               max_steps=1024, regen_env_at_steps=1024)
    env = LocalFrameRPYParamsEnv(cfg)
    obs, _ = env.vector_reset()
    acts = [np.random.rand(4) for _ in range(n)]
    for _ in range(5):
        env.vector_step(acts)
    t0 = time.perf_counter(); K = 200
    for _ in range(K):
        o, r, d, tr, info = env.vector_step(acts)
    dt = time.perf_counter() - t0
    print("list API, %d envs: %.1f us/step, %.3e env-steps/s" % (n, dt / K * 1e6, n * K / dt))
