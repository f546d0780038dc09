"""diagnostic (not a test): the per-step kernels of a T-step fragment captured once in a HIP graph (torch.cuda.CUDAGraph on the
stream qd_step launches on) and replayed -- the GPU-bound rate of one-kernel-per-step stepping, without the host launch path"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
n, T = 4096, 512
env, _ = bench.make_env("config3", n, 42, "cuda:0")
env.vector_reset_tensor()
acts = torch.rand((T, n, 4), device="cuda")
obs = torch.empty((T, n, env._dev.D), device="cuda"); rew = torch.empty((T, n), device="cuda"); tr = torch.empty((T, n), dtype=torch.uint8, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for t in range(64):
        env._dev.step(acts[t], obs[t], rew[t], tr[t])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for t in range(T):
            env._dev.step(acts[t], obs[t], rew[t], tr[t])
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 40
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("graph replay: %.3f us/step, %.1f M env-steps/s" % (dt / (reps * T) * 1e6, n * reps * T / dt / 1e6))
t0 = time.perf_counter()
for r in range(reps):
    for t in range(T):
        env._dev.step(acts[t], obs[t], rew[t], tr[t])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("direct launches: %.3f us/step, %.1f M env-steps/s" % (dt / (reps * T) * 1e6, n * reps * T / dt / 1e6))
# the productised form: qd_step_fragment through the C ABI
obs2 = torch.empty_like(obs)
env._dev.step_fragment(acts, obs2, rew, tr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    env._dev.step_fragment(acts, obs2, rew, tr)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("qd_step_fragment: %.3f us/step, %.1f M env-steps/s" % (dt / (reps * T) * 1e6, n * reps * T / dt / 1e6))
with torch.cuda.stream(s):
    env._dev.step_fragment(acts, obs2, rew, tr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        env._dev.step_fragment(acts, obs2, rew, tr)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("qd_step_fragment from a side stream: %.3f us/step, %.1f M env-steps/s" % (dt / (reps * T) * 1e6, n * reps * T / dt / 1e6))
# GPU-side duration of one fragment
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); env._dev.step_fragment(acts, obs2, rew, tr); e1.record(); torch.cuda.synchronize()
print("one fragment, event-timed on the caller's stream: %.3f us/step" % (e0.elapsed_time(e1) * 1000 / T))
