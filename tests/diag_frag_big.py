import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
for n in (131072,):
    e1, _ = bench.make_env("config3", n, 42, "cuda:0"); e2, _ = bench.make_env("config3", n, 42, "cuda:0")
    e1.vector_reset_tensor(); e2.vector_reset_tensor()
    T = 6
    acts = torch.rand((T, n, 4), device="cuda")
    obs, rew, tr = torch.empty((T, n, 22), device="cuda"), torch.empty((T, n), device="cuda"), torch.empty((T, n), dtype=torch.uint8, device="cuda")
    for rep in range(2):
        e1.step_fragment_tensor(acts, obs, rew, tr)
        for t in range(T):
            o, r, trn = e2.vector_step_tensor(acts[t])
            assert torch.allclose(obs[t], o, atol=1e-6) and torch.allclose(rew[t], r, atol=1e-6) and torch.equal(tr[t], trn), (rep, t)
    print("fragment == steps at", n, "envs (256-thread kernels)")
