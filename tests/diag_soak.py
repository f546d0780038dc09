"""diagnostic (not a test): long-run soak of the per-step path -- BASELINE config 3, 4096 envs, U[0,1) actions, in-kernel
auto-reset -- with invariants checked every `chunk` steps: finite observations / rewards, unit quaternions, bounded
positions (truncation keeps every env within max_distance of the reference), episode lengths, steady throughput.
usage: python tests/diag_soak.py [total_steps] [chunk]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
n = 4096
env, _ = bench.make_env("config3", n, 42, "cuda:0")
obs = env.vector_reset_tensor()
a = torch.rand((64, n, 4), device="cuda")
trunc_total = torch.zeros((), dtype=torch.int64, device="cuda")
print("steps        Msteps/s   truncations/step   max|pos-ref|   max|quat norm-1|   obs finite", flush=True)
done = 0
t_all = time.perf_counter()
while done < total:
    t0 = time.perf_counter()
    cnt = torch.zeros((), dtype=torch.int64, device="cuda")
    for k in range(chunk):
        o, r, tr = env.vector_step_tensor(a[k & 63])
        if (k & 1023) == 0:
            cnt += tr.sum()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    done += chunk
    q, v, act, sens, steps = env._dev.get_state()
    pos = q[:, :3].cpu().numpy(); quat = q[:, 3:7].cpu().numpy()
    ref = np.array([0, 0, 15.0])
    dist = np.linalg.norm(pos - ref, axis=1).max()
    qn = np.abs(np.linalg.norm(quat, axis=1) - 1).max()
    fin = bool(torch.isfinite(o).all() and torch.isfinite(r).all())
    sampled = chunk // 1024 + (1 if chunk % 1024 else 0)
    print("%-12d %-10.1f %-18.2f %-14.3f %-18.2e %s" % (done, n * chunk / dt / 1e6, float(cnt) / sampled, dist, qn, fin), flush=True)
    assert fin and dist < 4.5 and qn < 1e-3
    assert int(steps.max()) <= 1024
print("soak ok: %d steps x %d envs = %.2e env-steps in %.1f s" % (done, n, done * n, time.perf_counter() - t_all))
