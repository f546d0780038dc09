"""Diagnostic (not a test): where one workgroup of the policy kernel spends its time, from a -DQD_STAMPS build.
usage: QD_LIB=tests/_build/libqd_diag.so python tests/diag_policy_stamps.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_drone_amd.policy import DevicePolicy, compile_program
from mujoco_drone_amd import _lib as L
PG = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "policy_vectors.npz"))
w = {k: PG["rma_full/" + k] for k in PG["rma_full_keys"]}
pol = DevicePolicy("RMA_full", w)
d, ops, blob = compile_program("RMA_full", w)
names = ["lds init"] + ["op%d kind%d %d->%d%s" % (i, o.kind, o.in_dim, o.out_dim, " (value)" if o.flags else "") for i, o in enumerate(ops)] + ["outputs"]
lib = L.lib()
for n in (16, 4096):
    obs = torch.randn((n, 22), device="cuda"); prev = torch.rand((n, 4), device="cuda")
    for want_value in (False, True):
        acc = []
        for rep in range(30):
            for _ in range(5):
                pol.forward(obs, prev, want_value=want_value)
            torch.cuda.synchronize()
            buf = (C.c_ulonglong * 64)()
            assert lib.qd_debug_read_pstamps(buf) == 0
            st = np.array(buf[:3 + len(ops)], dtype=np.int64)
            acc.append(st)
        st = np.median(np.array(acc) - np.array(acc)[:, :1], axis=0)
        print("n=%d want_value=%s  total %.2f us" % (n, want_value, st[-1] / 100.0))
        prev_t = 0
        for k, name in enumerate(names):
            t = st[k + 1]
            if t >= prev_t and t > 0:
                print("   %-34s +%.2f us" % (name, (t - prev_t) / 100.0)); prev_t = t
