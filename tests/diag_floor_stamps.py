"""Diagnostic (not a test): cycle stamps of k_step_floor's workgroup 0 with every env resting on the floor (-DQD_STAMPS build).
usage: QD_LIB=tests/_build/libqd_stamps.so python tests/diag_floor_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mujoco_drone_amd import _lib as L  # noqa: E402
from mujoco_drone_amd.environments import _device as dev  # noqa: E402
from test_gpu_parity import make_cfg, rand_raw  # noqa: E402

n = 4096
rng = np.random.default_rng(0)
names = ["substep start", "forward done", "published + parked", "barrier passed", "solve done (wave 0)", "barrier passed",
         "collected", "integrated", "solve: contacts counted", "solve: mass matrix in LDS", "solve: tran done", "solve: first cost",
         "solve: Newton done"]
for load in (False, True):
    c = make_cfg(L, n, load=load, obs="BaseDroneEnv", reward="default_reward_fcn", frame_skip=1, h=0.002, ctrl_map=0, max_steps=10 ** 7, max_distance=1e9)
    c.floor_contact = 1
    env = dev.DeviceEnv(c)
    env.set_params(rand_raw(rng, n, load))
    nq, nv = (9, 8) if load else (7, 6)
    qpos = np.zeros((n, nq)); qpos[:, 3] = 1
    qpos[:, 2] = 0.02 if not load else 0.3
    if load:
        qpos[:, 7] = 1.2
    env.set_state(qpos, np.zeros((n, nv)), np.zeros((n, 4)))
    a = torch.zeros((n, 4), device="cuda")
    for _ in range(300):
        env.step(a)
    acc = []
    buf = (C.c_ulonglong * 64)()
    for _ in range(40):
        env.step(a)
        torch.cuda.synchronize()
        assert env.lib.qd_debug_read_sfstamps(buf) == 0
        st = np.array(buf[:13], dtype=np.int64)
        acc.append(st - st[0])
    med = np.median(np.array(acc), axis=0)
    order = [0, 1, 2, 3, 8, 9, 10, 11, 12, 4, 5, 6, 7]
    print("%s model, every env on the floor: median cycles since the substep started (workgroup 0, wave 0)" % ("load" if load else "single-body"))
    for k in order:
        print("  %-28s %8.0f" % (names[k], med[k]))
