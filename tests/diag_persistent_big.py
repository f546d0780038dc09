"""Diagnostic (not a test): the persistent fragment kernel against the per-step launches over the env-count range.
usage: QD_PERSISTENT_MAX_ENVS=1000000000 python tests/diag_persistent_big.py [envs,...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mujoco_drone_amd import _lib as L  # noqa: E402
from mujoco_drone_amd import parallel as par  # noqa: E402

envs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4096,16384,32768,65536,262144,1048576,4194304").split(",")]
for n in envs:
    T = 256 if n <= 65536 else (64 if n <= 1048576 else 16)
    for persistent in (1, 0):
        env, _alg = bench.make_env(os.environ.get("QD_DIAG_CONFIG", "config3"), n, 7, "cuda:0")
        env.vector_reset_tensor()
        env._dev.set_option(L.OPT_PERSISTENT_FRAGMENTS, persistent)
        f = par.FragmentBuffers(T, n, env._dev.D, "cuda:0")
        f.actions.copy_(torch.rand(f.actions.shape, device="cuda"))
        for _ in range(int(os.environ.get("QD_DIAG_WARM_STEPS", "0")) // T):      # into the steady state of truncations and resets
            env._dev.step_fragment(f.actions, f.obs, f.rewards, f.truncated)
        p, k = bench.kernel_period_us(env, f, launches=max(4 * T, 1024 if n <= 65536 else 0))
        if os.environ.get("QD_DIAG_WARM_STEPS"):
            print("   truncations per step per 64 envs in the last fragment: %.3f" % (float(f.truncated.sum()) / T / (n / 64)))
        print("n=%8d T=%4d %-28s %9.3f us per step = %.3e env-steps/s = %5.1f %% of 8 TB/s at SURVEY 8d's bytes" % (
            n, T, env._dev.fragment_kernel_name(), p, n / p * 1e6, bench.ALG_BYTES[_alg] * n / (p * 1e-6) / 8e12 * 100), flush=True)
        del env, f
        torch.cuda.empty_cache()
