"""Diagnostic (not a test): step-kernel throughput versus batch size (config 3)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import os
SIZES = [int(x) for x in os.environ.get('QD_SWEEP', '4096,16384,65536,262144,1048576,4194304').split(',')]
for n in SIZES:
    env, alg = bench.make_env("config3", n, 7, "cuda:0")
    env.vector_reset_tensor()
    a = torch.rand((4, n, 4), device="cuda")
    for i in range(30):
        env._dev.step(a[i % 4])
    iters = 400 if n <= 65536 else 60
    p = bench.stream_rate_us(env, a, launches=iters)
    print("envs %8d  period %9.2f us  env-steps/s %.3e  alg GB/s %7.1f  frac %.3f" % (n, p, n / (p * 1e-6), 309 * n / (p * 1e-6) / 1e9, 309 * n / (p * 1e-6) / 1e9 / 8000), flush=True)
    del env, a
    torch.cuda.empty_cache()
