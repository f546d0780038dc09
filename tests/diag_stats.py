"""diagnostic (not a test): streaming rate of the train-batch statistics kernels on BASELINE-sized fragments"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_drone_amd.custom_logging import BatchStatistics, EpisodeStatistics
T, N = 1024, 4096
st = BatchStatistics()
def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps
for name, cols in (("obs D=22", 22), ("obs D=23", 23), ("actions", 4), ("D=64", 64), ("D=37", 37)):
    x = torch.randn((T, N, cols), device="cuda")
    dt = timed(lambda: st.column_stats_tensor(x))
    print("column stats %-9s [%d x %d]: %.1f us, %.0f GB/s (%.1f %% of 8 TB/s)" % (name, T * N, cols, dt * 1e6, x.numel() * 4 / dt / 1e9, x.numel() * 4 / dt / 8e12 * 100), flush=True)
    del x
for n in (4096, 65536):
    rew = torch.randn((T, n), device="cuda"); tr = (torch.rand((T, n), device="cuda") < 0.001).to(torch.uint8)
    es = EpisodeStatistics(n)
    dt = timed(lambda: es.update_tensor(rew, tr))
    print("episode stats [%d x %d]: %.1f us, %.0f GB/s" % (T, n, dt * 1e6, T * n * 5 / dt / 1e9), flush=True)
# kernel-level split (main pass vs final fold) from the profiler: run under rocprofv3 --kernel-trace --stats
