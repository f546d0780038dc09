"""Diagnostic (not a test): the HIP runtime's launch-path knobs against the step loop -- graph-replayed fragments and per-step
calls of BASELINE config 3 -- one child process per setting.

Round 1 ran an uncommitted version of this script; its sixth child hung until the 300 s limit and the record (gpurun_out/knobs.txt)
does not say which setting that was.  This version cannot lose that information: the setting is printed and flushed BEFORE its
child starts, every child has its own (short) time limit and is ended by its exact PID, and a setting that times out is
reported as such and the sweep STOPS there (a GPU command killed at its limit has told you something: no further GPU step in
the same call).  Settings beyond the four round 1 got through are opt-in (QD_KNOBS_EXTRA=1): nothing in the library depends on
any of them.
"""
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys, time
sys.path.insert(0, os.path.dirname(%r))
import torch, bench
env, _ = bench.make_env("config3", 4096, 42, "cuda:0")
env.vector_reset_tensor()
T, n, D = 1024, 4096, 22
a = torch.rand((T, n, 4), device="cuda"); o = torch.empty((T, n, D), device="cuda"); r = torch.empty((T, n), device="cuda"); t = torch.empty((T, n), dtype=torch.uint8, device="cuda")
for _ in range(4):
    env._dev.step_fragment(a, o, r, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    env._dev.step_fragment(a, o, r, t)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / (4 * T)
s = env._dev.step
av = [a[i] for i in range(64)]
for i in range(1000): s(av[i & 63])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(4000): s(av[i & 63])
torch.cuda.synchronize()
print("graph %%.3f us/step   per-step calls %%.3f us/step" %% (dt * 1e6, (time.perf_counter() - t0) / 4000 * 1e6))
''' % HERE

SETTINGS = [{}, {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1"}, {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}, {"HIP_FORCE_DEV_KERNARG": "1"},
            {"HIP_FORCE_DEV_KERNARG": "0"}]
if os.environ.get("QD_KNOBS_EXTRA") == "1":   # candidates for round 1's unrecorded sixth setting; run at your own risk, one call each
    SETTINGS += [{"GPU_MAX_HW_QUEUES": "1"}, {"AMD_DIRECT_DISPATCH": "0"}, {"HIP_LAUNCH_BLOCKING": "1"}]
LIMIT = float(os.environ.get("QD_KNOBS_LIMIT_S", "90"))

for kn in SETTINGS:
    print("%-46s " % (kn or "(defaults)"), end="", flush=True)     # the record of WHAT runs comes before it runs
    p = subprocess.Popen([sys.executable, "-c", CHILD], env=dict(os.environ, **kn), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        out, _ = p.communicate(timeout=LIMIT)
        lines = [ln for ln in out.splitlines() if ln.startswith("graph")]
        print(lines[-1] if lines else "FAILED rc=%d: %s" % (p.returncode, out.strip().splitlines()[-1:] or ""), flush=True)
    except subprocess.TimeoutExpired:
        p.kill()
        p.wait()
        print("TIMED OUT after %.0f s -- this is the setting under which the loop does not finish; stopping the sweep" % LIMIT, flush=True)
        sys.exit(3)
