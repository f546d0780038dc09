"""Diagnostic (not a test): what rocprofv3 adds to a replayed graph of dependent launches.  512 launches of a near-empty kernel
(k_pid_reset: 64 workgroups zeroing three floats per env) are captured in one graph and replayed; the script prints the live
period per launch.  Run once plainly and once under `rocprofv3 --kernel-trace --stats` (tools/profile_r02.sh calib): the profiler's
average dispatch duration of a kernel that does nothing is its own share of every figure it reports for the step kernel."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

env, _ = bench.make_env("config3", 4096, 7, "cuda:0")
env.vector_reset_tensor()
dev = env._dev
T = 512
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    for _ in range(8):
        dev.pid_reset()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(T):
            dev.pid_reset()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    for it in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("near-empty kernel (k_pid_reset, 64 workgroups), graph of %d dependent launches: %.3f us per launch (live)" % (T, dt / (20 * T) * 1e6), flush=True)
