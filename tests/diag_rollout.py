"""Diagnostic (not a test): multi-step rollout kernel throughput."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for n in (4096, 16384):
    env, _ = bench.make_env("config3", n, 11, "cuda:0")
    env.vector_reset_tensor()
    a = torch.rand((256, n, 4), device="cuda")
    out = env._dev.rollout(a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8):
        env._dev.rollout(a, *out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("rollout kernel, %d envs: %.2f us/step, %.3e env-steps/s" % (n, dt / (8 * 256) * 1e6, 8 * 256 * n / dt))
