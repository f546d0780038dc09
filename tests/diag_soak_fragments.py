"""diagnostic (not a test): long-run soak of the persistent fragment kernels (k_rollout_lat) -- BASELINE config 3 at 4096 envs and
config 5 at 8192, 1024-step fragments of U[0,1) actions with the regen rule, in-kernel resets from the workgroup's own pool -- with
invariants checked every `every` fragments: finite rows / rewards, unit quaternions, every env within max_distance of its reference,
episode lengths, every reset served by the pool, no in-kernel poll ran out.   usage: python tests/diag_soak_fragments.py [fragments] [every]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

frags = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 200
T = 1024
for conf, n in (("config3", 4096), ("config5", 8192)):
    env, _ = bench.make_env(conf, n, 42, "cuda:0")
    env.vector_reset_tensor()
    D = env._dev.D
    acts = torch.rand((T, n, 4), device="cuda")
    O = torch.empty((T, n, D), device="cuda"); R = torch.empty((T, n), device="cuda"); Tr = torch.empty((T, n), dtype=torch.uint8, device="cuda")
    name = env._dev.fragment_kernel_name()
    trunc = 0
    t0 = time.perf_counter()
    for f in range(frags):
        env.step_fragment_tensor(acts, O, R, Tr)
        if (f + 1) % every == 0:
            torch.cuda.synchronize()
            fin = bool(torch.isfinite(O).all() and torch.isfinite(R).all())
            q, v, a, s, steps = env._dev.get_state()
            quat = q[:, 3:7]
            qn = float((quat.norm(dim=1) - 1).abs().max())
            trunc += int(Tr.sum())
            served, inline = env._dev.pool_counters()
            health = env._dev.health_counters()
            print("%-8s %-24s fragment %5d: finite %s  max |quat norm - 1| %.1e  max episode step %d  truncations in this fragment %d  resets served by the pool %d, sampled inline %d  health %s"
                  % (conf, name, f + 1, fin, qn, int(steps.max()), int(Tr.sum()), served, inline, health), flush=True)
            assert fin and qn < 1e-3 and int(steps.max()) <= 1024 and health == (0,)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s: %d fragments x %d steps x %d envs = %.2e env-steps in %.1f s (%.2e /s incl. checks)" % (conf, frags, T, n, frags * T * n, dt, frags * T * n / dt), flush=True)
print("fragment soak ok")
