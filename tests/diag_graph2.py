"""diagnostic: qd_step_fragment vs torch.cuda.CUDAGraph replay of the same launches, alternating"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
n, T, reps = 4096, int(os.environ.get("FRAG_T", "512")), int(os.environ.get("FRAG_REPS", "40"))
env, _ = bench.make_env("config3", n, 42, "cuda:0")
env.vector_reset_tensor()
acts = torch.rand((T, n, 4), device="cuda")
mk = lambda: (torch.empty((T, n, env._dev.D), device="cuda"), torch.empty((T, n), device="cuda"), torch.empty((T, n), dtype=torch.uint8, device="cuda"))
o1, r1, t1 = mk(); o2, r2, t2 = mk()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for t in range(64):
        env._dev.step(acts[t], o1[t], r1[t], t1[t])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for t in range(T):
            env._dev.step(acts[t], o1[t], r1[t], t1[t])
MODE = os.environ.get("FRAG_ON_SIDE", "0") == "1"
if MODE:
    with torch.cuda.stream(s):
        env._dev.step_fragment(acts, o2, r2, t2)
else:
    env._dev.step_fragment(acts, o2, r2, t2)
torch.cuda.synchronize()
def tg():
    with torch.cuda.stream(s):
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * T) * 1e6
def tf():
    if MODE:
        with torch.cuda.stream(s):
            t0 = time.perf_counter()
            for _ in range(reps):
                env._dev.step_fragment(acts, o2, r2, t2)
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (reps * T) * 1e6
    t0 = time.perf_counter()
    for _ in range(reps):
        env._dev.step_fragment(acts, o2, r2, t2)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * T) * 1e6
for k in range(4):
    print("round %d: qd_step_fragment %.3f us/step   torch graph %.3f us/step" % (k, tf(), tg()), flush=True)
