"""Diagnostic (not a test): per-phase cycle shares of the step kernel from a -DQD_STAMPS build.
usage: QD_LIB=tests/_build/libqd_diag.so [QD_DIAG_CONFIG=config5 QD_DIAG_ENVS=8192] python tests/diag_stamps.py
(build: python mujoco-drone_amd/build.py --variant diag -DQD_STAMPS)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench  # noqa: E402

CFG, N = os.environ.get("QD_DIAG_CONFIG", "config3"), int(os.environ.get("QD_DIAG_ENVS", 4096))
env, _ = bench.make_env(CFG, N, 42, "cuda:0")
env.vector_reset_tensor()
a = torch.rand((8, N, 4), device="cuda")
for i in range(300):
    env._dev.step(a[i % 8])
torch.cuda.synchronize()
lib = env._dev.lib
buf = (C.c_ulonglong * 512)()
acc = []
racc = []
for rep in range(50):
    for i in range(20):
        env._dev.step(a[i % 8])
    torch.cuda.synchronize()
    assert lib.qd_debug_read_stamps(buf) == 0
    st = np.array(buf[:], dtype=np.int64).reshape(64, 8)
    acc.append(np.diff(st, axis=1))
    rb = (C.c_ulonglong * 128)()
    assert lib.qd_debug_read_rstamps(rb) == 0
    rt = np.array(rb[:], dtype=np.int64).reshape(64, 2)
    racc.append(np.concatenate([rt[:, 1] - rt[:, 0], [rt[:, 1].max() - rt[:, 0].min()], [rt[:, 0].max() - rt[:, 0].min()]]))
acc = np.array(acc)  # [rep, wave, 7]
names = ["loads issued+arrived", "physics substep", "state/reward/trunc", "obs -> LDS", "state stores issued",
         "obs flush (LDS->global)", "store drain"]
med = np.median(acc.reshape(-1, 7), axis=0)
print("per-wave median cycles per phase (s_memtime ticks), total %d" % med.sum())
for n, m in zip(names, med):
    print("  %-28s %8.0f  %5.1f%%" % (n, m, 100 * m / med.sum()))
racc = np.array(racc)
print("realtime (100 MHz ticks): per-wave stamped region median %.1f ticks = %.2f us; first-start..last-end %.2f us; start skew %.2f us"
      % (np.median(racc[:, :64]), np.median(racc[:, :64]) / 100.0, np.median(racc[:, 64]) / 100.0, np.median(racc[:, 65]) / 100.0))
print("=> shader clock during the kernel ~ %.0f MHz" % (med.sum() / (np.median(racc[:, :64]) / 100.0)))
one = acc[-1]
tot = one.sum(axis=1)
order = np.argsort(tot)
print("per-wave totals (cycles), sorted:", tot[order].tolist())
print("slowest wave phases:", one[order[-1]].tolist(), "fastest:", one[order[0]].tolist())
rt0 = rt[:, 0] - rt[:, 0].min(); rt1 = rt[:, 1] - rt[:, 0].min()
print("start ticks:", rt0.tolist())
print("end ticks:", rt1.tolist())
# host-side cost of one step call (enqueue only): issue 3000 steps, time the Python loop, then the drain
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3000):
    env.vector_step_tensor(a[i % 8])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f us/step; total incl. drain %.2f us/step" % ((t1 - t0) / 3000 * 1e6, (t2 - t0) / 3000 * 1e6))
