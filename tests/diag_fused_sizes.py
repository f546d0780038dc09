"""diagnostic (not a test): where the fused closed loop stops paying -- RMA_full on config 3's env at growing batches, fused kernel
(QD_FUSED_MAX_ENVS raised so that it runs at every size) against two launches per step (QD_POLICY_UNFUSED=1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mujoco_drone_amd.policy import DevicePolicy, random_weights
mode = "two-launch" if os.environ.get("QD_POLICY_UNFUSED") else "fused"
for n in (16384, 32768, 65536, 131072):
    env, _ = bench.make_env("config3", n, 42, "cuda:0")
    pol = DevicePolicy("RMA_full", random_weights("RMA_full", 3))
    o = env.vector_reset_tensor().clone()
    pol.rollout(env._dev, 64, o)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pol.rollout(env._dev, 256, o)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-10s %7d envs  %.2f us/step (%.3e env-steps/s)" % (mode, n, dt / 256 * 1e6, n * 256 / dt), flush=True)
