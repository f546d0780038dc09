"""diagnostic (not a test): closed-loop policy rollout per family at 4096 envs; run once as is (fused kernel where one exists) and once
with QD_POLICY_UNFUSED=1 in the environment (two launches per step)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mujoco_drone_amd.policy import DevicePolicy, random_weights
n = 4096
mode = "two-launch" if os.environ.get("QD_POLICY_UNFUSED") else "fused"
for fam in ("RMA_full", "RMA_model", "RMA_model_smaller", "SimpleMLPmodel", "CustomMLP"):
    env, _ = bench.make_env("config3", n, 42, "cuda:0")
    pol = DevicePolicy(fam, random_weights(fam, 3))
    o = env.vector_reset_tensor().clone()
    for _ in range(2):
        pol.rollout(env._dev, 256, o)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2):
        pol.rollout(env._dev, 1024, o)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    print("%-20s %-10s %.2f us/step (%.3e env-steps/s)" % (fam, mode, dt / 1024 * 1e6, n * 1024 / dt), flush=True)
# train_LSTM.py's pair: CNNestimator on the config-5 env (23-value rows with the accelerometer: env phase behind the network)
for n5 in (4096, 8192):
    env, _ = bench.make_env("config5", n5, 42, "cuda:0")
    pol = DevicePolicy("CNNestimator", random_weights("CNNestimator", 3), obs_dim=23, num_states=23)
    o = env.vector_reset_tensor().clone()
    for _ in range(2):
        pol.rollout(env._dev, 256, o)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2):
        pol.rollout(env._dev, 1024, o)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    print("%-20s %-10s %.2f us/step (%.3e env-steps/s) at %d envs" % ("CNNestimator", mode, dt / 1024 * 1e6, n5 * 1024 / dt, n5), flush=True)
