"""Diagnostic (not a test): the per-step loop of an RL sampler whose policy is somebody else's kernels -- vector_step_tensor, then
torch ops that READ the observation rows and produce the next actions (here a 22x4 linear map + sigmoid).  Shows what the
non-temporal stores of the rows cost or save when a consumer reads them straight away."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

n = int(os.environ.get("QD_DIAG_ENVS", "4096"))
env, _ = bench.make_env("config3", n, 7, "cuda:0")
obs = env.vector_reset_tensor()
W = torch.randn((22, 4), device="cuda") * 0.1
act = torch.sigmoid(obs @ W)
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 3000
    for _ in range(K):
        obs, rew, tr = env.vector_step_tensor(act)
        act = torch.sigmoid(obs @ W)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("n=%d: %.2f us per (step + torch policy) iteration, %.3e env-steps/s" % (n, dt / K * 1e6, n * K / dt), flush=True)
# the same as one captured graph (no host in the loop): the GPU-side cost of the chain
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        obs, rew, tr = env.vector_step_tensor(act)
        act.copy_(torch.sigmoid(obs @ W))
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(64):
            obs, rew, tr = env.vector_step_tensor(act)
            act.copy_(torch.sigmoid(obs @ W))
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("n=%d graph-replayed chain: %.2f us per (step + torch policy) iteration" % (n, dt / (50 * 64) * 1e6), flush=True)
