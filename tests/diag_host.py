"""Diagnostic (not a test): where the host-side time of one step call goes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
env, _ = bench.make_env("config3", 4096, 42, "cuda:0")
env.vector_reset_tensor()
a = torch.rand((8, 4096, 4), device="cuda")
dev = env._dev
lib = dev.lib
N = 20000
def t(f, n=N):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): f(i)
    dt = time.perf_counter() - t0; torch.cuda.synchronize(); return dt / n * 1e6
print("empty loop            %.2f us" % t(lambda i: None))
print("qd_version (ctypes)   %.2f us" % t(lambda i: lib.qd_version()))
ap = a[0].data_ptr()
print("qd_step early-return  %.2f us" % t(lambda i: lib.qd_step(dev.handle, ap, 1, dev._obs_ptr, dev._rew_ptr, dev._trunc_ptr, 0)))
from mujoco_drone_amd.environments._device import _raw_stream
print("raw stream lookup     %.2f us" % t(lambda i: _raw_stream(0)))
print("a[i%%8] index          %.2f us" % t(lambda i: a[i % 8]))
x = a[0]
print("data_ptr+numel        %.2f us" % t(lambda i: (x.data_ptr(), x.numel())))
print("dev.step              %.2f us" % t(lambda i: dev.step(x), 5000))
print("vector_step_tensor    %.2f us" % t(lambda i: env.vector_step_tensor(x), 5000))
s = torch.cuda.current_stream().cuda_stream
print("raw qd_step full      %.2f us" % t(lambda i: lib.qd_step(dev.handle, ap, 16384, dev._obs_ptr, dev._rew_ptr, dev._trunc_ptr, s), 5000))
# pure host launch cost: a tiny kernel (k_set_ref, N=64) through the same ctypes path
from mujoco_drone_amd import _lib as L
import ctypes as C
c = L.QdConfig(); c.num_envs, c.model, c.obs_kind, c.reward_kind, c.frame_skip, c.max_steps, c.ctrl_map = 64, 1, 8, 2, 1, 512, 1
c.timestep, c.max_distance, c.per_env_reference = 0.01, 4.0, 1
from mujoco_drone_amd.environments._device import DeviceEnv
small = DeviceEnv(c)
r = torch.zeros((64, 4), device="cuda")
rp = r.data_ptr()
print("tiny-kernel launch loop   %.2f us" % t(lambda i: lib.qd_set_reference_per_env(small.handle, rp, s), 20000))
x64 = torch.rand((64, 4), device="cuda"); xp = x64.data_ptr()
print("qd_step N=64 loop         %.2f us" % t(lambda i: lib.qd_step(small.handle, xp, 256, small._obs_ptr, small._rew_ptr, small._trunc_ptr, s), 5000))
