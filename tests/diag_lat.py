"""Diagnostic (not a test): k_rollout_lat / k_rollout_coop at BASELINE config 3 -- time per step (plain library) or the four waves'
timeline of one step (a -DQD_STAMPS build in QD_LIB), for the configuration as it is and with the truncations taken away
(no in-kernel reset, no sampler job), to tell the step itself from what resets cost.
usage: [QD_LIB=tests/_build/libqd_stamps.so] python tests/diag_lat.py [envs] [T] [normal|noreset|nopool] [lat|coop] [config3|config5]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_drone_amd import _lib as QL  # noqa: E402
from mujoco_drone_amd import parallel as par  # noqa: E402
from mujoco_drone_amd.environments import observation_wrappers as ow, rewards  # noqa: E402
from mujoco_drone_amd.environments.BaseDroneEnv import base_config  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
mode = sys.argv[3] if len(sys.argv) > 3 else "normal"
kern = sys.argv[4] if len(sys.argv) > 4 else "lat"
cfg = dict(base_config)
cfg.update(num_drones=N, random_params=True, param_difficulty=1, state_difficulty=0.2, max_steps=1024, regen_env_at_steps=10 ** 9,
           reward_fcn=rewards.distance_energy_reward, seed=42, device="cuda:0", auto_reset=True)
if mode == "noreset":
    cfg.update(max_steps=10 ** 9, max_distance=1e9)
if mode == "nopool":
    cfg.update(random_start_pos=False)
if len(sys.argv) > 5 and sys.argv[5] == "config5":   # train_LSTM.py's configuration: sensor-reading rows, pendulum-energy reward, circle waypoints
    cfg.update(random_params=False, state_difficulty=0.8, reward_fcn=rewards.distance_energy_reward_pendulum_en4,
               reference_trajectory={"type": "circle", "radius": 1.0, "frequency": 0.5})
    env = ow.LocalFrameFullStateEnv(cfg)
else:
    env = ow.LocalFrameRPYParamsEnv(cfg)
if kern == "coop":
    env._dev.set_option(QL.OPT_LATENCY_KERNEL, 0)
env.vector_reset_tensor()
f = par.FragmentBuffers(T, N, env._dev.D, "cuda:0")
f.actions.copy_(torch.rand(f.actions.shape, device="cuda"))
lib = env._dev.lib
name = env._dev.fragment_kernel_name()
stamped = hasattr(lib, "qd_debug_read_rlstamps")
for _ in range(3):
    env.step_fragment_tensor(f.actions, f.obs, f.rewards, f.truncated)
torch.cuda.synchronize()
if not stamped:
    tot, reps = 0.0, 8
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        env.step_fragment_tensor(f.actions, f.obs, f.rewards, f.truncated)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    print("%s %s n=%d T=%d: %.3f us per step, %.2f truncations per step per 64 envs" % (
        name, mode, N, T, tot / reps * 1e3 / T, float(f.truncated.float().mean()) * 64), flush=True)
else:
    buf = (C.c_ulonglong * (64 * 4 * 16))()
    reader = lib.qd_debug_read_rlstamps if "k_rollout_lat" in name else lib.qd_debug_read_rcstamps
    acc = []
    for rep in range(24):
        env.step_fragment_tensor(f.actions, f.obs, f.rewards, f.truncated)
        torch.cuda.synchronize()
        assert reader(buf) == 0
        st = np.array(buf[:], dtype=np.int64).reshape(64, 4, 16)[:, :, :10]
        acc.append(st - st[:, :1, :1])
    acc = np.array(acc).reshape(-1, 4, 10)
    med = np.median(acc, axis=0)
    print("%s %s, step T/2 of a %d-step fragment, %d envs: median cycles since wave A's step start (waves A / B / C / D)" % (name, mode, T, N))
    for k, nm in enumerate(["step start (after barrier 2)", "phase 1 done", "barrier 1 passed", "phase 2 done", "barrier 2 passed"]):
        print("  %-30s %7.0f %7.0f %7.0f %7.0f" % (nm, med[0, k], med[1, k], med[2, k], med[3, k]))
    per = acc[:, 0, 4] - acc[:, 0, 0]
    print("  step period (wave A): median %.0f  p10 %.0f  p90 %.0f; phase 2 of wave A: median %.0f p10 %.0f p90 %.0f" % (
        np.median(per), np.percentile(per, 10), np.percentile(per, 90), *np.percentile(acc[:, 0, 3] - acc[:, 0, 2], [50, 10, 90])), flush=True)
    if "k_rollout_lat" in name:
        a = acc[:, 0, :]
        for nm, lo, hi in (("barrier 1 -> truncation known", 2, 5), ("reset block", 5, 6), ("reset block -> phase 2 done (pool take)", 6, 3), ("phase 2 done -> barrier 2", 3, 4)):
            d = a[:, hi] - a[:, lo]
            print("  wave A %-42s median %5.0f  p10 %5.0f  p90 %5.0f  mean %5.0f" % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90), d.mean()))
        for w, wn in ((1, "B"), (3, "D")):
            d = acc[:, w, 3] - acc[:, w, 2]
            print("  wave %s phase 2: median %5.0f p10 %5.0f p90 %5.0f mean %5.0f" % (wn, np.median(d), np.percentile(d, 10), np.percentile(d, 90), d.mean()))
        d = acc[:, 3, :]
        if d[:, 7].max() > 0:
            for nm, lo, hi in (("step start -> state in registers", 0, 8), ("-> phase 1 done", 8, 1), ("barrier 1 -> reward stored", 2, 7), ("-> rows stored (phase 2 done)", 7, 3)):
                x = d[:, hi] - d[:, lo]
                print("  wave D %-42s median %5.0f  p10 %5.0f  p90 %5.0f" % (nm, np.median(x), np.percentile(x, 10), np.percentile(x, 90)))
        print("  step period mean %.0f" % per.mean())
