"""Diagnostic (not a test): the three-wave cooperative step against the single-wave step -- per-launch period and the largest
difference of their outputs over 300 steps of BASELINE config 3 (run once per setting of QD_COOP_MAX_ENVS, compared on files)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mujoco_drone_amd import parallel as par  # noqa: E402

cfgname = os.environ.get("QD_DIAG_CONFIG", "config3")
n = int(os.environ.get("QD_DIAG_ENVS", "4096"))
tag = os.environ.get("QD_COOP_MAX_ENVS", "default")
noreset = os.environ.get("QD_DIAG_NORESET") == "1"
env, _ = bench.make_env(cfgname, n, 42, "cuda:0", auto_reset=not noreset)
env.vector_reset_tensor()
D = env._dev.D
g = torch.Generator(device="cuda"); g.manual_seed(7)
T = 300
acts = torch.rand((T, n, 4), generator=g, device="cuda")
obs = torch.empty((T, n, D), device="cuda"); rew = torch.empty((T, n), device="cuda"); tr = torch.empty((T, n), dtype=torch.uint8, device="cuda")
for t in range(T):
    env._dev.step(acts[t], obs[t], rew[t], tr[t])
torch.cuda.synchronize()
qpos, qvel, act, sens, steps = env._dev.get_state()
out = os.path.join("/tmp", "coop_%s_%s_%d_%d.npz" % (cfgname, tag, n, noreset))   # large: stays on the box
np.savez(out, obs=obs.cpu().numpy(), rew=rew.cpu().numpy(), tr=tr.cpu().numpy(), qpos=qpos.cpu().numpy(), qvel=qvel.cpu().numpy(),
         act=act.cpu().numpy(), sens=sens.cpu().numpy(), steps=steps.cpu().numpy())
f = par.FragmentBuffers(1024, n, D, "cuda:0")
f.actions.copy_(torch.rand(f.actions.shape, generator=g, device="cuda"))
for _ in range(3):
    p, k = bench.kernel_period_us(env, f)
    print("QD_COOP_MAX_ENVS=%s noreset=%d %s n=%d: %.3f us per launch (%d launches), truncations per step %.1f" % (tag, noreset, cfgname, n, p, k, float(f.truncated.float().sum()) / 1024), flush=True)
other = os.path.join("/tmp", "coop_%s_%s_%d_%d.npz" % (cfgname, "0" if tag != "0" else "default", n, noreset))
if os.path.exists(other):
    a, b = np.load(out), np.load(other)
    same_tr = bool((a["tr"] == b["tr"]).all())
    print("vs %s: truncation flags identical: %s" % (other, same_tr))
    for k in ("obs", "rew", "qpos", "qvel", "act", "sens"):
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
        print("  %-5s max |diff| %.3e   (max |value| %.3e)" % (k, d.max(), np.abs(b[k]).max()))
    print("  steps equal:", bool((a["steps"] == b["steps"]).all()))
