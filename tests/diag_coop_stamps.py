"""Diagnostic (not a test): timeline of the three waves of k_step_coop from a -DQD_STAMPS build.
usage: QD_LIB=tests/_build/libqd_diag.so python tests/diag_coop_stamps.py
(build: python mujoco-drone_amd/build.py --variant diag -DQD_STAMPS)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

CFG, N = os.environ.get("QD_DIAG_CONFIG", "config3"), int(os.environ.get("QD_DIAG_ENVS", 4096))
env, _ = bench.make_env(CFG, N, 42, "cuda:0", auto_reset=os.environ.get("QD_DIAG_NORESET") != "1")
env.vector_reset_tensor()
a = torch.rand((8, N, 4), device="cuda")
for i in range(300):
    env._dev.step(a[i % 8])
torch.cuda.synchronize()
lib = env._dev.lib
buf = (C.c_ulonglong * (64 * 3 * 16))()
acc = []
for rep in range(50):
    for i in range(20):
        env._dev.step(a[i % 8])
    torch.cuda.synchronize()
    assert lib.qd_debug_read_cstamps(buf) == 0
    st = np.array(buf[:], dtype=np.int64).reshape(64, 3, 16)[:, :, :10]
    acc.append(st - st[:, :, :1].min(axis=1, keepdims=True))      # relative to the workgroup's first wave start
acc = np.array(acc).reshape(-1, 3, 10)
med = np.median(acc, axis=0)
names = ["start", "loads arrived", "phase 1 done", "barrier 1 passed", "phase 2 done", "barrier 2 passed", "phase 3 done",
         "barrier 3 passed", "flush issued", "stores drained"]
print("median cycles since the workgroup's first wave started (waves A / B / C):")
for k, nm in enumerate(names):
    print("  %-18s %7.0f %7.0f %7.0f" % (nm, med[0, k], med[1, k], med[2, k]))
one = acc[-64:]
tot = one[:, :, 9].max(axis=1)
print("slowest workgroup of the last launch:", one[np.argmax(tot)].astype(int).tolist())
