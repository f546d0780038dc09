"""Diagnostic (not a test): where one pass of the fused policy -> env loop (k_rollout_fused_pipe) goes, from a -DQD_STAMPS build.
usage: QD_LIB=tests/_build/libqd_stamps.so python tests/diag_fused_stamps.py [envs] [cnn]
Stamps are s_memrealtime (10 ns) of thread 0 of workgroup 0, the last step of the fragment."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mujoco_drone_amd import _lib as L
from mujoco_drone_amd.policy import DevicePolicy, random_weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
device = torch.device("cuda:0")
if len(sys.argv) > 2 and sys.argv[2] == "cnn":     # train_LSTM.py's pair: 23-value rows (accelerometer + activations), CNNestimator
    env, _ = bench.make_env("config5", n, 0, device)
    pol = DevicePolicy("CNNestimator", random_weights("CNNestimator", 3), obs_dim=23, num_states=23, device=device)
else:
    env, _ = bench.make_env("config3", n, 0, device)
    pol = DevicePolicy("RMA_full", random_weights("RMA_full", 3), device=device)
o = env.vector_reset_tensor().clone()
lib = L.lib()
pol.rollout(env._dev, 64, o)
torch.cuda.synchronize()
wall = []
for rep in range(20):
    t0 = time.perf_counter()
    pol.rollout(env._dev, 256, o)
    torch.cuda.synchronize()
    wall.append(time.perf_counter() - t0)
print("closed policy loop (%s), %d envs: %.2f us per step (wall, 256-step fragment, median of 20)" % (pol.family, n, np.median(wall) / 256 * 1e6))
if hasattr(lib, "qd_debug_read_fpstamps"):
    buf = (C.c_ulonglong * 64)()
    assert lib.qd_debug_read_fpstamps(buf) == 0
    st = np.array(buf[:], dtype=np.int64)
    b = st[0]
    us = lambda k: (st[k] - b) / 100.0
    print("   last step, us since the pass started (thread 0 / lane 0 of each env wave):")
    print("   network : gather loads issued %.2f, past G0 %.2f, stored %.2f, past G1 %.2f, layers done %.2f, outputs start %.2f, done %.2f, past O %.2f"
          % (us(4), us(5), us(6), us(1), us(2), us(7), us(8), us(3)))
    print("   wave A  : stage 1 %.2f .. %.2f, stage 2 done %.2f, after O %.2f" % (us(16), us(17), us(18), us(19)))
    print("   wave D  : reward of the step before %.2f .. %.2f, row %.2f .. %.2f" % (us(42), us(43), us(40), us(41)))
