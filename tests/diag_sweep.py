"""Diagnostic (not a test): per-launch period of the step kernel over batch sizes (graph-replayed 256-step fragments)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mujoco_drone_amd import parallel as par  # noqa: E402
cfg = os.environ.get("QD_DIAG_CONFIG", "config3")
alg = {"config3": 309, "config5": 329, "config2": 181}[cfg]
for n in [int(x) for x in os.environ.get("QD_DIAG_SIZES", "4096,16384,65536,262144,1048576").split(",")]:
    env, _ = bench.make_env(cfg, n, 7, "cuda:0")
    (env.reset() if cfg == "config2" else env.vector_reset_tensor())
    T = 256 if n <= 65536 else 128
    f = par.FragmentBuffers(T, n, env._dev.D, "cuda:0")
    lo = 0.5 if cfg == "config2" else 0.0
    f.actions.copy_(lo + (1 - lo) * torch.rand(f.actions.shape, device="cuda"))
    p, k = bench.kernel_period_us(env, f, launches=4 * T)
    print("%s n=%8d: %9.3f us per launch  %.3e env-steps/s  %7.1f GB/s algorithmic = %.1f %% of 8 TB/s" %
          (cfg, n, p, n / (p * 1e-6), alg * n / (p * 1e-6) / 1e9, alg * n / (p * 1e-6) / 1e9 / 80), flush=True)
    del env, f
    torch.cuda.empty_cache()
