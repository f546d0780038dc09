"""Diagnostic (not a test): per-launch period of the step kernel against the fragment length T -- the observation / reward rows of
a fragment are written once each to T different places, so T sets how much streamed output passes between two visits of the
same state planes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mujoco_drone_amd import parallel as par  # noqa: E402
cfg = os.environ.get("QD_DIAG_CONFIG", "config3")
n = int(os.environ.get("QD_DIAG_ENVS", "4096"))
for T in [int(x) for x in os.environ.get("QD_DIAG_T", "1,4,16,64,256,1024,4096").split(",")]:
    env, _ = bench.make_env(cfg, n, 7, "cuda:0")
    env.vector_reset_tensor()
    f = par.FragmentBuffers(T, n, env._dev.D, "cuda:0")
    f.actions.copy_(torch.rand(f.actions.shape, device="cuda"))
    p, k = bench.kernel_period_us(env, f, launches=max(4096, 2 * T))
    print("%s n=%d T=%5d (%7.1f MB of rows per fragment): %.3f us per launch (%d launches)" % (cfg, n, T, T * n * (env._dev.D + 2) * 4 / 1e6, p, k), flush=True)
    del env, f
    torch.cuda.empty_cache()
