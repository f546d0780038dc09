"""diagnostic (not a test): how do the 64-thread and 256-thread k_step variants differ on the same envs?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_drone_amd import _lib as L
from mujoco_drone_amd.environments import _device as dev
from test_gpu_parity import make_cfg
for n in (98304,):
    mk = lambda k: dev.DeviceEnv(make_cfg(L, k, load=True, start=1, random_params=1, auto_reset=1, max_steps=7, seed=9))
    big, small = mk(n), mk(64)
    big.reset(); small.reset()
    for nm, x, y in zip(("qpos", "qvel", "act", "sens", "k"), big.get_state(), small.get_state()):
        print("after reset", nm, float((x[:64].double() - y.double()).abs().max()))
    g = torch.Generator(device="cuda").manual_seed(n)
    for t in range(10):
        a = torch.rand((n, 4), generator=g, device="cuda")
        ob, rb, tb = big.step(a)
        os_, rs, ts = small.step(a[:64].contiguous())
        d = (ob[:64] - os_).abs()
        print(t, "obs max diff %.3e" % float(d.max()), "cols", torch.nonzero(d.max(dim=0).values > 0).flatten().tolist(),
              "rew %.3e" % float((rb[:64] - rs).abs().max()), "trunc eq", bool(torch.equal(tb[:64], ts)))
        for nm, x, y in zip(("qpos", "qvel", "act", "sens"), big.get_state(), small.get_state()):
            dd = (x[:64].double() - y.double()).abs()
            if float(dd.max()) > 0:
                print("    ", nm, "%.3e" % float(dd.max()), torch.nonzero(dd.max(dim=0).values > 0).flatten().tolist())
