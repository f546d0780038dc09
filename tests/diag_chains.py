"""diagnostic: do two long kernels (k_rollout, T steps in one launch) from two streams of one process overlap?  And graphs?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mujoco_drone_amd import _lib as L
from mujoco_drone_amd.environments import _device as dev
from test_gpu_parity import make_cfg
n, T = 2048, 1024
for k in (1, 2, 4):
    envs, bufs, streams = [], [], [torch.cuda.Stream() for _ in range(k)]
    for j in range(k):
        with torch.cuda.stream(streams[j]):
            e = dev.DeviceEnv(make_cfg(L, n, load=True, start=1, random_params=1, auto_reset=1, max_steps=1024, seed=42 + j))
            e.reset()
            a = torch.rand((T, n, 4), device="cuda")
            o = torch.empty((T, n, e.D), device="cuda"); r = torch.empty((T, n), device="cuda"); t = torch.empty((T, n), dtype=torch.uint8, device="cuda")
            envs.append(e); bufs.append((a, o, r, t))
    torch.cuda.synchronize()
    for mode in ("rollout", "fragment"):
        def run():
            for j in range(k):
                with torch.cuda.stream(streams[j]):
                    (envs[j].rollout if mode == "rollout" else envs[j].step_fragment)(*bufs[j])
        run(); run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
        print("k=%d x %d envs, %-8s: %.2f ms per round (%.2f us per step per chain if serial: %.2f)" % (k, n, mode, dt * 1e3, dt / T * 1e6, dt / T / k * 1e6), flush=True)
