"""diagnostic (not a test): timing of the policy forward kernel and of the closed policy -> env loop"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_drone_amd.policy import DevicePolicy
from mujoco_drone_amd.environments.BaseDroneEnv import base_config
from mujoco_drone_amd.environments.observation_wrappers import LocalFrameRPYParamsEnv
from mujoco_drone_amd.environments.rewards import distance_energy_reward

PG = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "policy_vectors.npz"))
for tag, fam in (("rma_full", "RMA_full"), ("rma_model", "RMA_model"), ("simple_mlp", "SimpleMLPmodel")):
    w = {k: PG[tag + "/" + k] for k in PG[tag + "_keys"]}
    pol = DevicePolicy(fam, w)
    for n in (4096, 16384, 65536):
        obs = torch.randn((n, 22), device="cuda"); prev = torch.rand((n, 4), device="cuda")
        out = torch.empty((n, 4), device="cuda")
        for _ in range(20):
            pol.forward(obs, prev, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            pol.forward(obs, prev, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 200
        print("%-14s n=%6d forward %.2f us  (%.1f GFLOP/s incl. value head)" % (fam, n, us, 0), flush=True)
wa = {k: PG["rma_adapt/" + k] for k in PG["rma_adapt_keys"]}
pa = DevicePolicy("RMA_full_adapt", wa)
for n in (4096, 16384):
    obs = torch.randn((n, 22), device="cuda"); prev = torch.rand((n, 4), device="cuda"); out = torch.empty((n, 4), device="cuda")
    pa.reset_state(n)
    for k in range(20):
        pa.forward(obs, prev, out=out, counter=k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(200):
        pa.forward(obs, prev, out=out, counter=20 + k)
    e1.record(); torch.cuda.synchronize()
    print("RMA_full_adapt n=%6d forward %.2f us (kernel %d)" % (n, e0.elapsed_time(e1) * 1000 / 200, pa.kernel), flush=True)
w = {k: PG["rma_full/" + k] for k in PG["rma_full_keys"]}
pol = DevicePolicy("RMA_full", w)
for n in (4096, 16384):
    cfg = dict(base_config, num_drones=n, reward_fcn=distance_energy_reward, random_params=True, param_difficulty=1,
               state_difficulty=0.2, max_steps=1024, auto_reset=True)
    env = LocalFrameRPYParamsEnv(cfg)
    o = env.vector_reset_tensor().clone()
    pol.rollout(env._dev, 64, o)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    T = 512
    out = pol.rollout(env._dev, T, o)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("closed loop RMA_full n=%d: %.2f us/step, %.3e env-steps/s" % (n, dt / T * 1e6, n * T / dt), flush=True)
    t0 = time.perf_counter()
    out = pol.rollout(env._dev, T, o, explore=True, seed=1, want_logp=True, want_value=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("   exploring + logp + value   n=%d: %.2f us/step, %.3e env-steps/s" % (n, dt / T * 1e6, n * T / dt), flush=True)
