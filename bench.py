#!/usr/bin/env python3
"""Benchmark of the env step hot path (BASELINE.json metric: env steps/s at 4096 envs/GPU).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one vector_step of every env of the rank's batch: ctrl map, frame_skip physics
substeps, truncation, reward, observation, in-kernel auto-reset, plus the parameter
regeneration + full reset every `regen_env_at_steps` steps -- all inside the timed region.
Workload at N=1: BASELINE config 3 (drone + hanging load, 4096 envs, domain randomisation;
train_RMA.py:66-75 settings); N>1 is config 4: the same per GPU, rank seeds 42+rank, outputs written in
place into [T=1024,N,...] trajectory fragments.  The envs are independent, so the timed region has no
collective; the per-fragment RCCL all-gather that concatenates trajectories for a central learner is
measured separately (alone and overlapped with stepping) and reported under config.trajectory_all_gather
(SURVEY.md 8e).  Actions are synthetic U[0,1) tensors already resident in HBM.  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALG_BYTES = {"load22": 309, "noload6": 181, "load23": 329}  # algorithmic bytes per env-step
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
WORKLOADS = {
    "config3": "BASELINE config 3: drone + hanging load, domain-randomised params, LocalFrameRPYParamsEnv obs (D=22), "
               "distance_energy_reward, max_steps=1024, regen every 1024 steps, in-kernel auto-reset",
    "config2": "BASELINE config 2: SimpleDrone (no load), fixed init, U[0.5,1) rotor actions, 2 substeps at 1 kHz",
    "config5": "BASELINE config 5: drone + load, LocalFrameFullStateEnv obs (D=23), distance_energy_reward_pendulum_en4, "
               "state_difficulty 0.8, per-env moving circle waypoint (r=1, f=0.5 Hz) generated in the step kernel",
}


def make_env(kind, n, seed, device, auto_reset=True):
    from mujoco_drone_amd.environments.BaseDroneEnv import base_config
    from mujoco_drone_amd.environments import observation_wrappers as ow, rewards
    if kind == "config3":
        cfg = dict(base_config)
        cfg.update(num_drones=n, random_params=True, param_difficulty=1, state_difficulty=0.2, max_steps=1024,
                   regen_env_at_steps=1024, reward_fcn=rewards.distance_energy_reward, seed=seed, device=device,
                   auto_reset=auto_reset)
        return ow.LocalFrameRPYParamsEnv(cfg), "load22"
    if kind == "config5":
        cfg = dict(base_config)
        cfg.update(num_drones=n, random_params=False, state_difficulty=0.8, max_steps=1024, seed=seed, device=device,
                   reward_fcn=rewards.distance_energy_reward_pendulum_en4, auto_reset=auto_reset,
                   reference_trajectory={"type": "circle", "radius": 1.0, "frequency": 0.5})
        return ow.LocalFrameFullStateEnv(cfg), "load23"
    if kind == "config2":
        from mujoco_drone_amd.environments.SimpleDrone import SimpleDrone
        from mujoco_drone_amd import _lib as L
        return SimpleDrone(num_drones=n, reference=[0, 0, 1], device=device, seed=seed, random_start=L.START_FIXED,
                           auto_reset=auto_reset), "noload6"
    raise ValueError(kind)


def cpu_share():
    """host threads this process may really use: cgroup quota if set, else affinity, capped at 16
    (the CPU share of a one-GPU box; an uncapped count oversubscribes a shared host)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return int(os.environ.get("QD_CPU_THREADS", min(n, 16)))


def measured_copy_gbps(device):
    """device-to-device copy bandwidth of this box (read + write bytes), the practical HBM ceiling"""
    import torch
    a = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=device)   # 1 GiB
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    gbps = 5 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b
    torch.cuda.empty_cache()
    return gbps


def cpu_baseline(seconds_target=12.0, threads=None):
    """the oracle (C float64 port of the same step) on the host cores, config 3 at 4096 envs"""
    import numpy as np
    from oracle import oracle as orc
    from mujoco_drone_amd import _lib as L
    n = 4096
    rng = np.random.default_rng(0)
    center = np.array([1, 0.17, 7, 0.01, 1.2, 0.3]); width = np.array([0.1, 0.02, 1, 0.0025, 0.2, 0.05])
    raw = center + rng.uniform(-1, 1, (n, 6)) * width
    b = orc.Batch(raw, True, L.OBS_KINDS.index("LocalFrameRPYParamsEnv"), L.REWARD_KINDS.index("distance_energy_reward"),
                  0.01, 1, 1, [0, 0, 15, 0], 4.0, 1024)
    b.qpos[:, 2] = 15.0
    cores = threads or cpu_share()
    acts = rng.uniform(0, 1, (8, n, 4))
    b.step(acts[0], threads=cores)
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < seconds_target:
        for k in range(8):
            b.step(acts[k], threads=cores)
        steps += 8
    dt = time.perf_counter() - t0
    return {"value": n * steps / dt, "unit": "env steps/s", "cores": cores, "kind": "port",
            "sample": "config 3 (load model, D=22 obs, distance_energy_reward), 4096 envs x %d steps, oracle/qd_oracle.c "
                      "float64, OpenMP over envs; the reference's MuJoCo path is not runnable (mujoco absent)" % steps}


def kernel_time_us(env, actions, samples=200):
    """average duration of ONE step-kernel launch, HIP events on the launch stream around single launches
    issued on an idle stream (minus the cost of an empty event pair)"""
    import torch
    s = torch.cuda.current_stream()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(samples)]
    em = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(samples)]
    step = env._dev.step
    for i in range(samples):
        torch.cuda.synchronize()
        ev[i][0].record(s)
        step(actions[i % actions.shape[0]])
        ev[i][1].record(s)
    torch.cuda.synchronize()
    for i in range(samples):
        torch.cuda.synchronize()
        em[i][0].record(s)
        em[i][1].record(s)
    torch.cuda.synchronize()
    import statistics
    t = statistics.median(a.elapsed_time(b) for a, b in ev) * 1e3
    e = statistics.median(a.elapsed_time(b) for a, b in em) * 1e3
    return max(t - e, 1e-3), t, e


def stream_rate_us(env, actions, launches=4000):
    """back-to-back launches, whole region between two events: per-launch period when the GPU queue is full"""
    import torch
    step = env._dev.step
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    P = actions.shape[0]
    torch.cuda.synchronize()
    a.record()
    for i in range(launches):
        step(actions[i % P])
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / launches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8192)
    ap.add_argument("--warmup", type=int, default=512)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--config", default="config3", choices=["config3", "config2", "config5"])
    ap.add_argument("--fragment", type=int, default=1024, help="steps per all-gathered trajectory fragment (N>1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from mujoco_drone_amd import parallel as par
    rank, world, local = par.init_distributed()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if os.environ.get("QD_SINGLE_DEVICE"):   # rehearsal of the N>1 path on a one-GPU box (with QD_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    device = "cuda:%d" % local
    n, K, W = args.envs, args.steps, args.warmup

    env, alg = make_env(args.config, n, par.shard_seed(42, rank), device)
    step = env.vector_step_tensor if args.config != "config2" else env.step_tensor
    if args.config != "config2":
        env.vector_reset_tensor()
        D = env._dev.D
    else:
        env.reset()
        D = 6
    lo, hi = (0.0, 1.0) if args.config != "config2" else (0.5, 1.0)
    P = 64
    g = torch.Generator(device=device); g.manual_seed(1000 + rank)
    actions = lo + (hi - lo) * torch.rand((P, n, 4), generator=g, device=device, dtype=torch.float32)

    T = min(args.fragment, K)
    # Rollout fragments [T,N,...] the step kernel writes in place; the synthetic actions live in the fragment's action
    # tensor itself -- where a policy would write them.  Every step is one k_step launch (qd_step); the T launches of a
    # fragment are enqueued as ONE HIP graph (qd_step_fragment: captured on first use, replayed afterwards), because at
    # 4096 envs the per-launch host path (4-5.5 us depending on the host CPU) is what bounds a step-by-step loop, not the
    # 4.1 us kernel.  Steps that do not fill a fragment go through the per-step call.  The envs never exchange data, so
    # the timed region has no collective; with N > 1 the per-fragment RCCL all-gather that hands trajectories
    # to a central learner is measured right after it, alone and overlapped with stepping, and reported separately.
    pending = [None, None]
    use_graph = not os.environ.get("QD_BENCH_NO_GRAPH")
    frags = [par.FragmentBuffers(T, n, D, device) for _ in range(2)]
    gathers = [par.FragmentGather(f, world) for f in frags] if world > 1 else None
    for f in frags:
        f.actions.copy_(lo + (hi - lo) * torch.rand(f.actions.shape, generator=g, device=device, dtype=torch.float32))
    state = {"cur": 0, "gathers": 0, "graph_steps": 0, "call_steps": 0, "use_graph": use_graph}
    # per-step slices of the fragments, made once (tensor indexing costs more than the launch it feeds)
    views = [[(f.actions[t], (f.obs[t], f.rewards[t], f.truncated[t])) for t in range(T)] for f in frags]

    def run(k_steps, base=0, gather=False):
        t = 0
        while t < k_steps:
            tt = (base + t) % T
            cur = state["cur"]
            if gather and tt == 0 and pending[cur] is not None:      # this buffer's previous gather must have drained
                for w in pending[cur]:
                    w.wait()
                pending[cur] = None
            f = frags[cur]
            if state["use_graph"] and tt == 0 and k_steps - t >= T:
                try:
                    env.step_fragment_tensor(f.actions, f.obs, f.rewards, f.truncated)
                except Exception as ex:      # a runtime that cannot capture: fall back to per-step calls for the rest of the run
                    state["use_graph"] = False
                    print("bench: HIP graph path disabled (%r)" % (ex,), file=sys.stderr)
                    continue
                state["graph_steps"] += T
                t += T
                tt = T - 1
            else:
                a_t, out_t = views[cur][tt]
                step(a_t, out=out_t)
                state["call_steps"] += 1
                t += 1
            if tt == T - 1:
                if gather:
                    pending[cur] = gathers[cur](f, async_op=True)[1]
                    state["gathers"] += 1
                state["cur"] = cur ^ 1

    def drain():
        for b in range(2):
            if pending[b] is not None:
                for w in pending[b]:
                    w.wait()
                pending[b] = None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # clock ramp: a fresh process finds the GPU in a low power state and a 4 us kernel every 5 us takes tens of ms to pull the
    # shader clock up (measured: the same K steps are 2-12 % slower after 512 untimed steps than after 8192).  These extra
    # untimed steps come before the W warmup steps of the contract and are reported in config.clock_ramp_steps.
    ramp = (8192 + T - 1) // T * T - W        # >= 8192 - W steps, and the timed region starts on a fragment boundary
    while ramp < 0:
        ramp += T
    ramp_s = float(os.environ.get("QD_BENCH_RAMP_S", "0"))
    if ramp_s > 0:
        t_r = time.perf_counter()
        while time.perf_counter() - t_r < ramp_s:
            run(8192, base=ramp)
            ramp += 8192
            torch.cuda.synchronize()
    else:
        run(ramp)
    run(W, base=ramp)
    fence()
    state["graph_steps"] = state["call_steps"] = 0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()           # HIP events on the stream the step kernels are launched on (torch's current stream)
    run(K, base=ramp + W)
    ev1.record()
    fence()
    dt = time.perf_counter() - t0
    timed_graph_steps, timed_call_steps = state["graph_steps"], state["call_steps"]
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    gather_info = None
    if world > 1:
        def maxed(x):
            t_ = torch.tensor([x], dtype=torch.float64, device=device)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            return float(t_.item())
        # (i) one fragment all-gather alone
        gathers[0](frags[0])
        fence()
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            gathers[0](frags[0])
        fence()
        gather_ms = maxed((time.perf_counter() - t1) / reps * 1e3)
        # (ii) stepping with the gathers overlapped (double-buffered fragments, asynchronous collective)
        ks = max(T, min(K, 4 * T) // T * T)
        state["cur"] = 0
        fence()
        t1 = time.perf_counter()
        run(ks, base=0, gather=True)
        drain()
        fence()
        overl = maxed(time.perf_counter() - t1)
        gather_info = {"all_gather_ms_per_fragment": gather_ms, "fragment_steps": T,
                       "all_gather_bytes_per_rank_per_fragment": frags[0].nbytes(),
                       "env_steps_per_sec_with_overlapped_all_gather": world * n * ks / overl,
                       "all_gather_algbw_GBps": world * frags[0].nbytes() / (gather_ms * 1e-3) / 1e9}

    out = None
    if rank == 0:
        value = world * n * K / dt
        out = {"metric": "env_steps_per_sec", "value": value, "unit": "env steps/s", "n_gpus": world, "steps": K,
               "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": WORKLOADS[args.config] + "; trajectories written in place into [T=%d,N,...] fragments" % T +
                                      (" (their RCCL all-gather is reported separately in config.trajectory_all_gather)" if world > 1 else ""),
                          "envs_per_gpu": n, "global_envs": world * n, "clock_ramp_steps": ramp, "frame_skip": 2 if args.config == "config2" else 1,
                          "launch": ("one k_step kernel launch per step (C ABI); %d of the %d timed steps enqueued as HIP graphs of %d launches "
                                     "(qd_step_fragment), %d through per-step qd_step calls" % (timed_graph_steps, K, T, timed_call_steps)),
                          "precision": "float32 state / trigonometry / drag / integration, float64 inertia assembly and solves (load model)", "parallelism": "env-sharded x%d" % world,
                          "trajectory_all_gather": gather_info}}
        # ---- roofline of the dominant kernel (k_step), measured live with HIP events -----------------
        # average launch duration of k_step over the timed region: HIP events on the launch stream bracket the K
        # back-to-back launches (the regen launches every 1024 steps are < 0.1 % of it), so elapsed / K is the
        # kernel's average duration including the inter-kernel boundary -- the same quantity rocprofv3's kernel
        # trace reports for back-to-back dispatches (profiles/r01_n4096_rocprof_summary.json)
        kus = ev0.elapsed_time(ev1) * 1e3 / K
        iso_us, raw_us, empty_us = kernel_time_us(env, actions)
        bytes_per_launch = ALG_BYTES[alg] * n
        achieved = bytes_per_launch / (kus * 1e-6) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(args.config, {}).get(str(n))
            except Exception:
                traffic = None
        copy_gbps = measured_copy_gbps(device)
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": "qd::k_step",
                           "kernel_us": kus, "isolated_launch_us": iso_us, "algorithmic_bytes_per_env_step": ALG_BYTES[alg],
                           "env_steps_per_launch": n, "measured_copy_GBps": copy_gbps,
                           "frac_of_measured_copy": achieved / copy_gbps, "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / "
                           "WRITE_SIZE passes of this command, FETCH_SIZE x2 per the gfx950 calibration)" if traffic else None,
                           "note": "4096 envs = 64 wavefronts on 1024 SIMDs: the launch is latency/occupancy-bound, "
                                   "not HBM-bound (see DESIGN.md and the env-count sweep in `extras`)"}
        if not args.no_extras and world == 1:
            extras = {}
            try:
                # the same K steps through the per-step Python API (vector_step_tensor): bound by the host launch path
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for t in range(K):
                    step(actions[t % P])
                torch.cuda.synchronize()
                extras["per_step_api_env_steps_per_s"] = n * K / (time.perf_counter() - t1)
                sweep = []
                for nn in (4096, 65536, 1048576, 4194304):
                    e2, alg2 = make_env(args.config, nn, 7, device)
                    (e2.vector_reset_tensor() if args.config != "config2" else e2.reset())
                    a2 = lo + (hi - lo) * torch.rand((4, nn, 4), device=device, dtype=torch.float32)
                    for _ in range(20):
                        e2._dev.step(a2[0])
                    p2 = stream_rate_us(e2, a2, launches=300 if nn <= 65536 else 60)
                    sweep.append({"envs": nn, "period_us": p2, "env_steps_per_s": nn / (p2 * 1e-6),
                                  "alg_GBps": ALG_BYTES[alg2] * nn / (p2 * 1e-6) / 1e9,
                                  "frac_hbm": ALG_BYTES[alg2] * nn / (p2 * 1e-6) / 1e9 / HBM_PEAK_GBS})
                    del e2, a2
                    torch.cuda.empty_cache()
                extras["env_count_sweep"] = sweep
                # multi-step kernel (state in registers across T steps)
                e3, _ = make_env(args.config, n, 11, device)
                if args.config != "config2":
                    e3.vector_reset_tensor()
                    a3 = torch.rand((256, n, 4), device=device, dtype=torch.float32)
                    e3._dev.rollout(a3)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(8):
                        e3._dev.rollout(a3)
                    torch.cuda.synchronize()
                    extras["rollout_kernel_env_steps_per_s"] = 8 * 256 * n / (time.perf_counter() - t1)
                    # closed loop with the on-device analytic PID cascade as the action source (attitude_test.py's loop)
                    e3._dev.rollout_pid(256)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(8):
                        e3._dev.rollout_pid(256)
                    torch.cuda.synchronize()
                    extras["pid_closed_loop_env_steps_per_s"] = 8 * 256 * n / (time.perf_counter() - t1)
                # what the reference logs about a train batch (custom_logging.py:9-31, training.py:16-22), over the fragment the
                # timed steps just wrote: per-column min / max / mean / var of obs and actions, episode returns / lengths
                from mujoco_drone_amd.custom_logging import BatchStatistics, EpisodeStatistics
                bs, es = BatchStatistics(), EpisodeStatistics(n, device)
                f0 = frags[0]
                for name, fn, nbytes in (("obs", lambda: bs.column_stats_tensor(f0.obs), f0.obs.numel() * 4),
                                         ("actions", lambda: bs.column_stats_tensor(f0.actions), f0.actions.numel() * 4),
                                         ("episodes", lambda: es.update_tensor(f0.rewards, f0.truncated), f0.rewards.numel() * 5)):
                    for _ in range(3):
                        fn()
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record()
                    for _ in range(20):
                        fn()
                    ev1.record()
                    torch.cuda.synchronize()
                    us = ev0.elapsed_time(ev1) * 1000.0 / 20
                    extras["fragment_%s_stats_us" % name] = us
                    extras["fragment_%s_stats_GBps" % name] = nbytes / (us * 1e-6) / 1e9
                if args.config == "config3":
                    # the reference's actor (RMA_full, train_PPO.py:39-45, random-init weights) inside the loop:
                    # policy forward (f32 MFMA) -> env step, 2 launches per step enqueued by one C call, nothing leaves the GPU
                    from mujoco_drone_amd.policy import DevicePolicy, random_weights
                    pol = DevicePolicy("RMA_full", random_weights("RMA_full", 3), device=device)
                    o3 = e3.vector_reset_tensor().clone()
                    pa = torch.empty((n, 4), device=device)
                    for _ in range(20):
                        pol.forward(o3, out=pa)
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record()
                    for _ in range(200):
                        pol.forward(o3, out=pa)
                    ev1.record()
                    torch.cuda.synchronize()
                    extras["policy_forward_us"] = ev0.elapsed_time(ev1) * 1000.0 / 200
                    extras["policy_forward_TFLOPs"] = 2 * 57792 * n / (extras["policy_forward_us"] * 1e-6) / 1e12
                    pol.rollout(e3._dev, 64, o3)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    pol.rollout(e3._dev, 1024, o3)
                    torch.cuda.synchronize()
                    extras["policy_closed_loop_env_steps_per_s"] = 1024 * n / (time.perf_counter() - t1)
                    t1 = time.perf_counter()
                    pol.rollout(e3._dev, 1024, o3, explore=True, seed=42, want_logp=True, want_value=True)
                    torch.cuda.synchronize()
                    extras["policy_sample_batch_env_steps_per_s"] = 1024 * n / (time.perf_counter() - t1)  # sampled actions + logp + value
                    extras["policy_kernel"] = "specialised" if pol.kernel > 0 else "interpreter"
                    # train_RMA.py's network: RMA_full with the adaptation CNN over the 32-step history (incremental, per-env rings)
                    pad = DevicePolicy("RMA_full_adapt", random_weights("RMA_full_adapt", 4), device=device)
                    pad.reset_state(n)
                    for k in range(20):
                        pad.forward(o3, out=pa, counter=k)
                    ev0.record()
                    for k in range(200):
                        pad.forward(o3, out=pa, counter=20 + k)
                    ev1.record()
                    torch.cuda.synchronize()
                    extras["adapt_policy_forward_us"] = ev0.elapsed_time(ev1) * 1000.0 / 200
                    pad.reset_state(n)
                    pad.rollout(e3._dev, 64, o3)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    pad.rollout(e3._dev, 1024, o3, counter0=64)
                    torch.cuda.synchronize()
                    extras["adapt_policy_closed_loop_env_steps_per_s"] = 1024 * n / (time.perf_counter() - t1)
                other = "config2" if args.config != "config2" else "config3"
                e4, alg4 = make_env(other, n, 5, device)
                (e4.vector_reset_tensor() if other == "config3" else e4.reset())
                lo4, hi4 = (0.0, 1.0) if other == "config3" else (0.5, 1.0)
                # like the headline: one kernel launch per step, the launches of a 1024-step fragment replayed from a HIP graph
                a4 = lo4 + (hi4 - lo4) * torch.rand((1024, n, 4), device=device, dtype=torch.float32)
                D4 = e4._dev.D
                o4 = torch.empty((1024, n, D4), device=device); r4 = torch.empty((1024, n), device=device)
                t4 = torch.empty((1024, n), dtype=torch.uint8, device=device)
                for _ in range(2):
                    e4.step_fragment_tensor(a4, o4, r4, t4)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(4):
                    e4.step_fragment_tensor(a4, o4, r4, t4)
                torch.cuda.synchronize()
                extras[other + "_env_steps_per_s"] = 4 * 1024 * n / (time.perf_counter() - t1)
            except Exception as ex:  # extras never invalidate the headline line
                extras["error"] = repr(ex)
            out["extras"] = extras
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
            one = cpu_baseline(seconds_target=3.0, threads=1)
            out["cpu_baseline"]["single_core_value"] = one["value"]
            out["cpu_baseline"]["reference_python_overhead_bound"] = (
                "the reference's own per-step Python (state extraction + obs + reward, physics excluded) measured in "
                "BASELINE.md section 2 caps it at <= 1.25e4 env steps/s per process")
        elif world > 1:
            out["cpu_baseline"] = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
