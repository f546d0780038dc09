"""Diagnostic (not a test): time per env step of qd_step_fragment for the training configuration -- ONE persistent launch
(k_rollout_coop) or, with QD_PERSISTENT=0 in the environment, T per-step launches replayed from a HIP graph.
usage: python tests/diag_persistent.py [envs,envs,...] [T,T,...] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mujoco_drone_amd import parallel as par  # noqa: E402

envs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4096").split(",")]
Ts = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1024").split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
mode = "per-step launches (graph)" if os.environ.get("QD_PERSISTENT") == "0" else "one persistent launch"
for n in envs:
    for T in Ts:
        env, _ = bench.make_env("config3", n, 7, "cuda:0")
        env.vector_reset_tensor()
        f = par.FragmentBuffers(T, n, env._dev.D, "cuda:0")
        f.actions.copy_(torch.rand(f.actions.shape, device="cuda"))
        for _ in range(3):
            env.step_fragment_tensor(f.actions, f.obs, f.rewards, f.truncated)
        torch.cuda.synchronize()
        best = 1e9
        tot = 0.0
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            env.step_fragment_tensor(f.actions, f.obs, f.rewards, f.truncated)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            best = min(best, ms)
            tot += ms
        us = tot / reps * 1e3 / T
        print("%s  n=%d T=%d: %.3f us per step (best %.3f) = %.3e env-steps/s" % (mode, n, T, us, best * 1e3 / T, n / us * 1e6), flush=True)
        del env, f
        torch.cuda.empty_cache()
