"""Diagnostic (not a test): the single-wave multi-step kernel k_rollout against per-step launches for BASELINE config 2 (SimpleDrone)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mujoco_drone_amd import parallel as par  # noqa: E402

for n in (4096, 65536, 1048576):
    T = 1024 if n <= 4096 else (256 if n <= 65536 else 32)
    env, _ = bench.make_env("config2", n, 7, "cuda:0")
    env.reset()
    f = par.FragmentBuffers(T, n, 6, "cuda:0")
    f.actions.copy_(0.5 + 0.5 * torch.rand(f.actions.shape, device="cuda"))
    p, k = bench.kernel_period_us(env, f, launches=4 * T)
    print("config2 n=%d T=%d %-24s %.3f us per step = %.3e env-steps/s = %.1f %% (181 B)" % (n, T, env._dev.fragment_kernel_name(), p, n / p * 1e6, 181 * n / (p * 1e-6) / 8e12 * 100), flush=True)
    for _ in range(2):
        env._dev.rollout(f.actions, f.obs, f.rewards, f.truncated)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        env._dev.rollout(f.actions, f.obs, f.rewards, f.truncated)
    e1.record()
    torch.cuda.synchronize()
    p = e0.elapsed_time(e1) * 1e3 / (4 * T)
    print("config2 n=%d T=%d %-24s %.3f us per step = %.3e env-steps/s = %.1f %% (181 B)" % (n, T, "k_rollout<false,64,3>", p, n / p * 1e6, 181 * n / (p * 1e-6) / 8e12 * 100), flush=True)
    del env, f
    torch.cuda.empty_cache()
