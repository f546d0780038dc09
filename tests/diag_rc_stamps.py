"""Diagnostic (not a test): timeline of the four waves of k_rollout_coop for one step in the middle of a fragment, from a
-DQD_STAMPS build.
usage: QD_LIB=tests/_build/libqd_stamps.so python tests/diag_rc_stamps.py [T] [pid]     (pid: the PID cascade as the action source)
(build: python mujoco-drone_amd/build.py --variant stamps -DQD_STAMPS)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mujoco_drone_amd import parallel as par  # noqa: E402

N, T = int(os.environ.get("QD_DIAG_ENVS", 4096)), int(sys.argv[1]) if len(sys.argv) > 1 else 256
env, _ = bench.make_env("config3", N, 42, "cuda:0")
if os.environ.get("QD_DIAG_LATENCY") == "0":
    from mujoco_drone_amd import _lib as QL
    env._dev.set_option(QL.OPT_LATENCY_KERNEL, 0)
env.vector_reset_tensor()
f = par.FragmentBuffers(T, N, env._dev.D, "cuda:0")
f.actions.copy_(torch.rand(f.actions.shape, device="cuda"))
lib = env._dev.lib
buf = (C.c_ulonglong * (64 * 4 * 16))()
acc = []
PID = len(sys.argv) > 2 and sys.argv[2] == "pid"
if PID:
    env.pid_reset()
for rep in range(24):
    if PID:
        env._dev.rollout_pid(T)
    else:
        env.step_fragment_tensor(f.actions, f.obs, f.rewards, f.truncated)
    torch.cuda.synchronize()
    reader = lib.qd_debug_read_rlstamps if "k_rollout_lat" in env._dev.fragment_kernel_name() else lib.qd_debug_read_rcstamps
    assert reader(buf) == 0
    st = np.array(buf[:], dtype=np.int64).reshape(64, 4, 16)[:, :, :5]
    acc.append(st - st[:, :1, :1])            # relative to wave A's start of the stamped step
acc = np.array(acc).reshape(-1, 4, 5)
med = np.median(acc, axis=0)
names = ["step start (after barrier 2)", "phase 1 done", "barrier 1 passed", "phase 2 done", "barrier 2 passed"]
print("%s%s, step T/2 of a %d-step fragment, %d envs: median cycles since wave A's step start (waves A / B / C / D)" % (env._dev.fragment_kernel_name(), " (PID cascade in wave B's phase 2)" if PID else "", T, N))
for k, nm in enumerate(names):
    print("  %-30s %7.0f %7.0f %7.0f %7.0f" % (nm, med[0, k], med[1, k], med[2, k], med[3, k]))
per = acc[:, 0, 4] - acc[:, 0, 0]
print("step period (wave A, barrier 2 to barrier 2): median %.0f  p10 %.0f  p90 %.0f cycles" % (np.median(per), np.percentile(per, 10), np.percentile(per, 90)))
