"""Diagnostic (not a test): what a short fragment's launch + synchronize costs on the host side, with the default wait mode and with
hipDeviceScheduleSpin (argv[1] == "spin": hipSetDeviceFlags(1) before the runtime touches the device)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
if mode == "spin":
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(hipDeviceScheduleSpin) ->", hip.hipSetDeviceFlags(1))
import numpy as np, torch, bench
env, _ = bench.make_env("config3", 4096, 0, torch.device("cuda:0"))
env.vector_reset_tensor()
dev = env._dev
for T in (20, 256):
    acts = torch.rand((T, 4096, 4), device="cuda")
    O = torch.empty((T, 4096, dev.D), device="cuda"); R = torch.empty((T, 4096), device="cuda"); Tr = torch.empty((T, 4096), dtype=torch.uint8, device="cuda")
    for _ in range(200):
        dev.step_fragment(acts, O, R, Tr)
    torch.cuda.synchronize()
    ts = []
    for _ in range(300):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev.step_fragment(acts, O, R, Tr)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print("%s: T=%d  launch + synchronize: median %.1f us  p10 %.1f  p90 %.1f  (kernel alone ~ %.1f us)" % (mode, T, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90), 1.35 * T))
