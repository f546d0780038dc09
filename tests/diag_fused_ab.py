"""diagnostic (not a test): the closed policy loop at a few (network, batch) points, for A/B comparisons of library variants on
ONE box: QD_LIB=tests/_build/libqd_<variant>.so python tests/diag_fused_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mujoco_drone_amd.policy import DevicePolicy, random_weights
for fam, conf, n, kw in (("RMA_full", "config3", 4096, {}), ("RMA_full", "config3", 8192, {}),
                         ("CNNestimator", "config5", 8192, dict(obs_dim=23, num_states=23)), ("CNNestimator", "config5", 16384, dict(obs_dim=23, num_states=23))):
    env, _ = bench.make_env(conf, n, 42, "cuda:0")
    pol = DevicePolicy(fam, random_weights(fam, 3), **kw)
    o = env.vector_reset_tensor().clone()
    for _ in range(2):
        pol.rollout(env._dev, 256, o)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pol.rollout(env._dev, 1024, o)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("%-14s %6d envs  %.2f us/step (%.3e env-steps/s)" % (fam, n, best / 1024 * 1e6, n * 1024 / best), flush=True)
