"""GPU tests of the persistent fragment kernel (k_rollout_coop, csrc/qd_rollout_coop.hip): qd_step_fragment / qd_rollout of the
training configuration (BASELINE config 3 / 4: load model, LocalFrameRPYParamsEnv, distance_energy_reward) at small batches are
ONE launch that keeps 64 envs per workgroup on a CU for all T steps.  Its contract is "the same as T x qd_step" -- the
per-step cooperative kernel is the comparison throughout (obs, reward, truncation flags step by step, the state and the
accelerometer plane left behind, the reset pool handed back), plus the float64 oracle at BASELINE size.

Reference path replaced: T consecutive BaseDroneEnv.vector_step calls (environments/BaseDroneEnv.py:259-294) as the sampler
issues them per 1024-step fragment (train_RMA.py:63)."""
import numpy as np
import pytest

from divergence import Divergence
from test_gpu_parity import make_cfg, qd  # noqa: F401  (qd is a fixture)

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _bufs(T, n, D=22):
    kw = dict(device="cuda")
    return (torch.empty((T, n, D), **kw), torch.empty((T, n), **kw), torch.empty((T, n), dtype=torch.uint8, **kw))


def _heading_safe_absdiff(a, b):
    d = (a - b).abs()
    d[..., 5] = torch.minimum(d[..., 5], (d[..., 5] - 2 * np.pi).abs())   # the heading error wraps at +-pi
    return d


# Two separately compiled kernels: the same functions on the same values, but the compiler fuses multiply-adds differently in
# the two contexts -- measured 4.8e-7 in an observation entry after ONE step (half an ulp of the 15 m altitude), as between
# the other launch variants of the step (tests/diag_variant_diff.py).  Episodes here are <= 12 steps and a
# reset re-synchronises an env exactly (the sample is a pure function of seed, env and episode), so the comparison to the
# per-step kernel is made at ten ulps of the altitude; what IS bit-exact -- the persistent kernel against itself across batch
# sizes and fragment cuts -- is asserted bit-exactly in test_fragment_is_batch_and_cut_invariant.
OBS_ATOL = REW_ATOL = STATE_ATOL = 2e-5


class _Worst:
    def __init__(self):
        self.obs = self.rew = self.state = 0.0

    def step(self, O, o, R, r, Tr, tr, what):
        assert torch.equal(Tr, tr), what
        do, dr = float(_heading_safe_absdiff(O, o).max()), float((R - r).abs().max())
        self.obs, self.rew = max(self.obs, do), max(self.rew, dr)
        assert do <= OBS_ATOL and dr <= REW_ATOL, "%s: |d obs| %.3e |d reward| %.3e" % (what, do, dr)

    def states(self, a, b, what):
        for x, y, name in zip(a.get_state(), b.get_state(), ("qpos", "qvel", "act", "sensordata", "num_steps")):
            if name == "num_steps":
                assert torch.equal(x, y), what
            else:
                tol = 2e-3 if name == "sensordata" else STATE_ATOL      # the accelerometer amplifies rate differences (x mass^-1)
                d = float((x - y).abs().max())
                self.state = max(self.state, d) if name != "sensordata" else self.state
                assert d <= tol, "%s %s: %.3e" % (what, name, d)

    def __str__(self):
        return "max |d obs| %.2e  |d reward| %.2e  |d state| %.2e" % (self.obs, self.rew, self.state)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 300, 4097, 40001])
def test_fragment_equals_per_step_kernel(qd, n):
    """ragged and multi-workgroup batch sizes, short episodes (every env resets several times inside the fragment), T = 1 and
    longer runs, fragments that continue each other: every output of every step and the state left behind equal the per-step
    cooperative kernel's (truncation flags exactly, values to the rounding of two compilations of the same arithmetic)"""
    L = qd._lib
    mk = lambda: qd.dev.DeviceEnv(make_cfg(L, n, load=True, start=1, random_params=1, auto_reset=1, max_steps=7, seed=9))
    a, b = mk(), mk()
    a.reset(); b.reset()
    g = torch.Generator(device="cuda").manual_seed(n)
    w = _Worst()
    for T in (1, 23, 8):
        acts = torch.rand((T, n, 4), generator=g, device="cuda")
        O, R, Tr = _bufs(T, n)
        a.step_fragment(acts, O, R, Tr)
        for t in range(T):
            o, r, tr = b.step(acts[t])
            w.step(O[t], o, R[t], r, Tr[t], tr, "T %d t %d" % (T, t))
        w.states(a, b, "after T = %d" % T)
    print("n = %d: fragment vs per-step kernel, %s" % (n, w))
    assert int(Tr.sum()) > 0


@pytest.mark.parametrize("n", [300, 4097, 40001])
def test_fragment_with_a_reset_in_every_step(qd, n):
    """max_steps = 1: every lane truncates and is reset in EVERY step, so the reward of step t - 1 (of the state before that
    step's reset, handed over in LDS) is due while the solver wave already publishes step t's pre-reset state.  Rewards, rows and
    flags against the per-step kernel, and twice in a row against itself (a race would make the launch irreproducible)."""
    L = qd._lib
    mk = lambda: qd.dev.DeviceEnv(make_cfg(L, n, load=True, start=1, random_params=1, auto_reset=1, max_steps=1, seed=3))
    a, b, c = mk(), mk(), mk()
    for e in (a, b, c):
        e.reset()
    T = 24
    acts = torch.rand((T, n, 4), generator=torch.Generator(device="cuda").manual_seed(n), device="cuda")
    O, R, Tr = _bufs(T, n)
    a.step_fragment(acts, O, R, Tr)
    O2, R2, Tr2 = _bufs(T, n)
    c.step_fragment(acts, O2, R2, Tr2)
    assert torch.equal(O, O2) and torch.equal(R, R2) and torch.equal(Tr, Tr2)
    w = _Worst()
    for t in range(T):
        o, r, tr = b.step(acts[t])
        w.step(O[t], o, R[t], r, Tr[t], tr, "t %d" % t)
    w.states(a, b, "after %d steps" % T)
    assert int(Tr.sum()) == T * n
    print("n = %d, a reset in every step: fragment vs per-step kernel, %s" % (n, w))


@pytest.mark.parametrize("N", [4097, 40001])
def test_fragment_is_batch_and_cut_invariant(qd, N):
    """what must hold bit for bit: env i's rows do not depend on the batch it is in (N envs vs the first 64 of them alone) nor
    on where a run is cut into fragments (40 steps at once vs 13 + 1 + 26), through in-kernel resets, including the state, the
    accelerometer plane and the episode counters left in the arena"""
    L, T = qd._lib, 40
    mk = lambda k: qd.dev.DeviceEnv(make_cfg(L, k, load=True, start=1, random_params=1, auto_reset=1, max_steps=7, seed=9))
    big, small, cut = mk(N), mk(64), mk(N)
    if N > 16384:   # bit-exactness holds within one kernel: above 16384 envs that is k_rollout_coop, so the 64-env batch runs it too
        small.set_option(L.OPT_LATENCY_KERNEL, 0)
    assert big.fragment_kernel_name() == small.fragment_kernel_name() == cut.fragment_kernel_name()
    for e in (big, small, cut):
        e.reset()
    acts = torch.rand((T, N, 4), device="cuda")
    Ob, Rb, Tb = _bufs(T, N)
    big.step_fragment(acts, Ob, Rb, Tb)
    Os, Rs, Ts = _bufs(T, 64)
    small.step_fragment(acts[:, :64].contiguous(), Os, Rs, Ts)
    assert torch.equal(Ob[:, :64], Os) and torch.equal(Rb[:, :64], Rs) and torch.equal(Tb[:, :64], Ts)
    Oc, Rc, Tc = _bufs(T, N)
    for lo, hi in ((0, 13), (13, 14), (14, 40)):
        cut.step_fragment(acts[lo:hi], Oc[lo:hi], Rc[lo:hi], Tc[lo:hi])
    assert torch.equal(Ob, Oc) and torch.equal(Rb, Rc) and torch.equal(Tb, Tc), "cuts: max |d obs| %.3e, first differing step %d" % (
        float((Ob - Oc).abs().max()), int(torch.nonzero((Ob != Oc).flatten(1).any(dim=1))[0]))
    for x, y in zip(big.get_state(), cut.get_state()):
        assert torch.equal(x, y)
    for x, y in zip(big.get_state(), small.get_state()):
        assert torch.equal(x[:64], y)
    names = qd.dev.ARENA_PLANES
    pb, ps = big.planes()[:, :64], small.planes()
    for k, name in enumerate(names):                              # everything the arena holds; the reset pool too while both batches
        if N > 16384 and name.startswith(("NX", "NY")):           # run the in-workgroup sampler (above 16384 envs lanes sample inline:
            continue                                              # same states, nothing left in the pool planes)
        assert torch.equal(pb[k], ps[k]), "arena plane %s" % name


def test_latency_and_throughput_kernels_agree(qd):
    """the two persistent kernels (k_rollout_lat up to 16384 envs, k_rollout_coop above, QD_OPT_LATENCY_KERNEL = 0 everywhere) run
    the same step in two arrangements of the same equations (qd_dynamics.h: mass_inverse / solve_inv5 against mass_factor /
    reduce_rhs / finish_accel): flags identical, values to rounding over short episodes, the state left behind likewise"""
    L, n, T = qd._lib, 3000, 40
    mk = lambda: qd.dev.DeviceEnv(make_cfg(L, n, load=True, start=1, random_params=1, auto_reset=1, max_steps=9, seed=4))
    a, b = mk(), mk()
    b.set_option(L.OPT_LATENCY_KERNEL, 0)
    assert "k_rollout_lat<1>" in a.fragment_kernel_name() and "k_rollout_coop<1" in b.fragment_kernel_name()
    a.reset(); b.reset()
    acts = torch.rand((T, n, 4), generator=torch.Generator(device="cuda").manual_seed(5), device="cuda")
    (Oa, Ra, Ta), (Ob, Rb, Tb) = _bufs(T, n), _bufs(T, n)
    a.step_fragment(acts, Oa, Ra, Ta)
    b.step_fragment(acts, Ob, Rb, Tb)
    w = _Worst()
    for t in range(T):
        w.step(Oa[t], Ob[t], Ra[t], Rb[t], Ta[t], Tb[t], "t %d" % t)
    w.states(a, b, "after %d steps" % T)
    print("k_rollout_lat vs k_rollout_coop: %s" % w)


def test_fragment_and_per_step_launches_interleave(qd):
    """the arena is the hand-over: fragment -> single steps -> rollout -> single steps give what single steps alone give, and the
    reset pool the fragment hands back serves the per-step kernels' resets (nothing sampled inline after the first episodes)"""
    L, n = qd._lib, 500
    mk = lambda: qd.dev.DeviceEnv(make_cfg(L, n, load=True, start=1, random_params=1, auto_reset=1, max_steps=9, seed=5))
    a, b = mk(), mk()
    a.reset(); b.reset()
    g = torch.Generator(device="cuda").manual_seed(3)
    w = _Worst()
    plan = [("frag", 20), ("step", 11), ("roll", 13), ("step", 9), ("frag", 30), ("step", 10)]
    for kind, T in plan:
        acts = torch.rand((T, n, 4), generator=g, device="cuda")
        O, R, Tr = _bufs(T, n)
        if kind == "frag":
            a.step_fragment(acts, O, R, Tr)
        elif kind == "roll":
            O, R, Tr = a.rollout(acts)
        else:
            for t in range(T):
                o, r, tr = a.step(acts[t])
                O[t], R[t], Tr[t] = o, r, tr
        for t in range(T):
            o, r, tr = b.step(acts[t])
            w.step(O[t], o, R[t], r, Tr[t], tr, "%s t %d" % (kind, t))
    w.states(a, b, "at the end")
    print("interleaved launches vs per-step kernel, %s" % w)
    taken, inline = a.pool_counters()
    print("interleaved: %d in-kernel resets from the pool, %d sampled inline" % (taken, inline))
    assert taken > 8 * n and inline <= 0.02 * taken


def test_fragment_without_auto_reset_and_with_fixed_start(qd):
    """auto_reset off: flags are reported, envs keep flying; fixed start (no pool): in-kernel resets go back to start_pos"""
    L, n, T = qd._lib, 200, 15
    for kw in (dict(start=1, auto_reset=0, max_steps=5), dict(start=0, auto_reset=1, max_steps=4)):
        mk = lambda: qd.dev.DeviceEnv(make_cfg(L, n, load=True, random_params=1, seed=2, **kw))
        a, b = mk(), mk()
        a.reset(); b.reset()
        acts = torch.rand((T, n, 4), device="cuda")
        O, R, Tr = _bufs(T, n)
        a.step_fragment(acts, O, R, Tr)
        w = _Worst()
        for t in range(T):
            o, r, tr = b.step(acts[t])
            if kw["auto_reset"]:
                w.step(O[t], o, R[t], r, Tr[t], tr, "%s t %d" % (kw, t))
            else:          # 15 steps without a re-synchronising reset: only the flags and a loose bound
                assert torch.equal(Tr[t], tr) and float(_heading_safe_absdiff(O[t], o).max()) < 1e-4, (kw, t)
        if kw["auto_reset"]:
            w.states(a, b, str(kw))
        assert int(Tr.sum()) > 0


def test_fragment_with_moving_and_per_env_references(qd):
    """the waypoint generators (evaluation.py:135-152) evaluated inside the persistent kernel: the reference of a step is the one
    of the episode step the step STARTED from, a reset row uses episode step 0; per-env static references come from the arena"""
    from mujoco_drone_amd.environments.BaseDroneEnv import base_config
    from mujoco_drone_amd.environments.observation_wrappers import LocalFrameRPYParamsEnv
    from mujoco_drone_amd.environments.rewards import distance_energy_reward
    n, T = 192, 40
    common = dict(base_config, num_drones=n, reward_fcn=distance_energy_reward, random_params=True, param_difficulty=1,
                  state_difficulty=0.2, max_steps=12, auto_reset=True)
    for extra in (dict(reference_trajectory=dict(type="circle", radius=0.5, frequency=0.5)), dict(per_env_reference=True)):
        cfg = dict(common, **extra)
        e1, e2 = LocalFrameRPYParamsEnv(cfg), LocalFrameRPYParamsEnv(cfg)
        if "per_env_reference" in extra:
            refs = torch.tensor([0.0, 0.0, 15.0, 0.0], device="cuda") + 0.3 * torch.randn((n, 4), device="cuda")
            e1.set_reference_tensor(refs); e2.set_reference_tensor(refs)
        e1.vector_reset_tensor(); e2.vector_reset_tensor()
        acts = torch.rand((T, n, 4), device="cuda")
        O, R, Tr = _bufs(T, n)
        e1.step_fragment_tensor(acts, O, R, Tr)
        w = _Worst()
        for t in range(T):
            o, r, tr = e2.vector_step_tensor(acts[t])
            w.step(O[t], o, R[t], r, Tr[t], tr, "%s t %d" % (extra, t))
        print("%s: %s" % (sorted(extra), w))


def test_config3_full_size_fragment_vs_oracle_200_steps(qd, orc):
    """BASELINE config 3 at its full size THROUGH THE PERSISTENT KERNEL: 4096 envs, domain-randomised parameters, random initial
    states, 200 steps of U[0,1) rotor actions as four 50-step fragments; every env's state against the float64 oracle at step
    200 (<= 1e-4, the BASELINE bar), every step's observation rows and rewards against the oracle's (measured maxima printed)"""
    rng = np.random.default_rng(123)
    n, L, steps, F = 4096, qd._lib, 200, 50
    c = make_cfg(L, n, load=True, start=1, random_params=1, seed=42, difficulty=1.0, sdiff=0.2, max_steps=10 ** 6, max_distance=1e9)
    env = qd.dev.DeviceEnv(c)
    env.reset()
    raw = env.get_params().cpu().numpy()
    q0, v0, a0, _, _ = [x.cpu().numpy().astype(np.float64) for x in env.get_state()]
    ob = orc.Batch(raw, True, L.OBS_KINDS.index("LocalFrameRPYParamsEnv"), L.REWARD_KINDS.index("distance_energy_reward"),
                   0.01, 1, 1, (0, 0, 15, 0), 1e9, 10 ** 6)
    ob.qpos[:], ob.qvel[:], ob.act[:] = q0, v0, a0
    acts = rng.uniform(0, 1, (steps, n, 4)).astype(np.float32)
    dacts = torch.as_tensor(acts).cuda()
    O, R, Tr = _bufs(F, n)
    worst_o = worst_r = 0.0
    for f in range(steps // F):
        env.step_fragment(dacts[f * F:(f + 1) * F].contiguous(), O, R, Tr)
        Oh, Rh = O.cpu().numpy().astype(np.float64), R.cpu().numpy().astype(np.float64)
        for t in range(F):
            oo, orr, otr = ob.step(acts[f * F + t].astype(np.float64), threads=8)
            d = np.abs(Oh[t] - oo)
            d[:, 5] = np.minimum(d[:, 5], np.abs(d[:, 5] - 2 * np.pi))
            worst_o, worst_r = max(worst_o, float(d.max())), max(worst_r, float(np.abs(Rh[t] - orr).max()))
        assert int(Tr.sum()) == 0
    gq, gv, ga, gs, gk = [x.cpu().numpy().astype(np.float64) for x in env.get_state()]
    assert np.all(gk == steps)
    div = Divergence(True)
    div.update(dict(qpos=gq, qvel=gv, act=ga), dict(qpos=ob.qpos, qvel=ob.qvel, act=ob.act))
    print(div.table("config 3 through k_rollout_coop at step 200, all 4096 envs"))
    print("max over 200 steps x 4096 envs: |obs - oracle| %.3e, |reward - oracle| %.3e" % (worst_o, worst_r))
    assert div.max("rel") < 1e-4 and div.max("mixed") < 1e-4
    assert worst_o < 1e-3 and worst_r < 1.25e-3          # 2 x the measured maxima (4.9e-4, 6.1e-4 on MI355X; positions of 15 m, d^2 of 16)
    np.testing.assert_allclose(gs, ob.sensor, rtol=2e-4, atol=2e-3)      # the last step's accelerometer reading (quirk C-6)


def test_fragment_long_run_invariants_and_pool(qd):
    """BASELINE-shaped run: 4096 envs, 1024-step fragments with the regen rule, three regen periods.  Every in-kernel reset of
    the fragments must have been served by the workgroup's own sampler (LDS pool), quaternions stay unit, everything finite,
    every env stays within max_distance of its reference after each step (truncation + reset work)"""
    import bench
    n, T = 4096, 1024
    env, _ = bench.make_env("config3", n, 42, "cuda:0")
    env.vector_reset_tensor()
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = torch.rand((T, n, 4), generator=g, device="cuda")
    O, R, Tr = _bufs(T, n)
    resets = 0
    for k in range(3):
        env.step_fragment_tensor(acts, O, R, Tr)
        resets += int(Tr[:-1].sum())
        assert bool(Tr[-1].all())                      # the regen boundary: everybody truncated (BaseDroneEnv.py:289-291)
        assert torch.isfinite(O).all() and torch.isfinite(R).all()
        assert float(O[:, :, :3].norm(dim=2).max()) <= 4.0 + 0.05    # e_l: a row is either inside the bound or a new episode's first
    taken, inline = env._dev.pool_counters()
    print("3 x 1024 steps of 4096 envs: %d in-kernel resets, %d from the LDS pool, %d sampled inline" % (resets, taken, inline))
    # (the regen rule overwrites the flags of each fragment's last step, where a few dozen envs were reset in the kernel as well)
    assert resets <= taken + inline <= resets + 3 * 100 and resets > 20000
    assert inline <= 0.01 * resets
    q = env._dev.get_state()[0]
    assert torch.allclose(q[:, 3:7].norm(dim=1), torch.ones(n, device=q.device), atol=1e-5)


def test_fragment_buffers_are_validated(qd):
    """qd_step_fragment writes through raw pointers: strided views, wrong dtypes and host tensors are refused in Python"""
    L, n, T = qd._lib, 64, 4
    env = qd.dev.DeviceEnv(make_cfg(L, n, load=True, start=1, random_params=1, auto_reset=1, max_steps=7, seed=9))
    env.reset()
    acts = torch.rand((T, n, 4), device="cuda")
    O, R, Tr = _bufs(T, n)
    env.step_fragment(acts, O, R, Tr)
    with pytest.raises(ValueError):
        env.step_fragment(acts, O, R, Tr.to(torch.float32))
    with pytest.raises(ValueError):
        env.step_fragment(acts, torch.empty((T, n, 44), device="cuda")[:, :, ::2], R, Tr)
    with pytest.raises(ValueError):
        env.step_fragment(acts, O, R.cpu(), Tr)
    with pytest.raises(ValueError):
        env.step_fragment(acts[:, :, :3], O, R, Tr)


# ------------------------------------------------------------------ the other instantiations: train_LSTM.py's configuration, run-time dispatch
def _close_rows(O, o, acc_at, what, tol_acc=3e-3, tol=None):
    """rows of a sensor-carrying variant: everything at OBS_ATOL except the three accelerometer entries (the reading amplifies
    rate differences by 1 / inertia; and a reset row's reading comes from the pool's affine form in the per-step kernel, from the
    next step's own solve here)"""
    d = _heading_safe_absdiff(O, o)
    if acc_at is not None:
        da = float(d[..., acc_at:acc_at + 3].max())
        d[..., acc_at:acc_at + 3] = 0
        assert da <= tol_acc, "%s: accelerometer entries %.3e" % (what, da)
    assert float(d.max()) <= (tol or 5 * OBS_ATOL), "%s: %.3e" % (what, float(d.max()))


@pytest.mark.parametrize("n", [500, 20000])
def test_config5_fragment_equals_per_step_kernel(qd, n):
    """BASELINE config 5 (train_LSTM.py: LocalFrameFullStateEnv, distance_energy_reward_pendulum_en4, circle waypoint per env)
    through k_rollout_coop<SPEC_LSTM>: the row carries the accelerometer, so rows are completed a round late and a reset row
    takes the next round's reading.  Against the single-wave per-step kernel (k_step<true,64,2>, monolithic forward dynamics):
    flags exactly, values to rounding; both register-budget instantiations (<= 16384 envs / above)"""
    import bench
    T = 48
    e1, _ = bench.make_env("config5", n, 42, "cuda:0")
    e2, _ = bench.make_env("config5", n, 42, "cuda:0")
    e1.vector_reset_tensor(); e2.vector_reset_tensor()
    assert ("k_rollout_lat<2>" if n <= 16384 else "k_rollout_coop<2") in e1._dev.fragment_kernel_name()
    g = torch.Generator(device="cuda").manual_seed(5)
    resets = 0
    for rep in range(2):
        # two float32 implementations of a chaotic system (state_difficulty 0.8, episodes of hundreds of steps, the monolithic
        # forward dynamics against its three pieces): they drift apart at ~1e-6 per step early on, 1.5e-4 after 67 steps.  Each
        # 48-step fragment therefore starts from the same arena; the bound that matters is the float64 oracle's (below)
        e2._dev.arena.copy_(e1._dev.arena)
        acts = torch.rand((T, n, 4), generator=g, device="cuda")
        O, R, Tr = _bufs(T, n, 23)
        e1.step_fragment_tensor(acts, O, R, Tr)
        for t in range(T):
            o, r, tr = e2.vector_step_tensor(acts[t])
            assert torch.equal(Tr[t], tr), (rep, t)
            _close_rows(O[t], o, 12, "rep %d t %d" % (rep, t), tol=3e-4)
            assert float((R[t] - r).abs().max()) <= 1e-3, (rep, t, float((R[t] - r).abs().max()))
        resets += int(Tr.sum())
        for x, y, name in zip(e1._dev.get_state(), e2._dev.get_state(), ("qpos", "qvel", "act", "sensordata", "num_steps")):
            tol = 0 if name == "num_steps" else (3e-3 if name == "sensordata" else 3e-4)
            assert float((x.float() - y.float()).abs().max()) <= tol, (name, float((x.float() - y.float()).abs().max()))
    assert resets > 0.002 * n * T          # state_difficulty 0.8: resets do happen inside 96 steps


def test_config5_latency_kernel_is_batch_and_cut_invariant(qd):
    """k_rollout_lat<SPEC_LSTM> (sensor-reading rows: the reading handed over a round late, reset lanes' entries filled in by the store
    wave, the half round at the end of every fragment): 40 steps at once == 13 + 1 + 26 bit for bit, and 4097 envs == the first 64
    alone (static reference: config 5's waypoint phase depends on the batch size), including the state and the accelerometer plane"""
    L, T, N = qd._lib, 40, 4097
    mk = lambda k: qd.dev.DeviceEnv(make_cfg(L, k, load=True, obs="LocalFrameFullStateEnv", reward="distance_energy_reward_pendulum_en4",
                                             start=1, random_params=0, auto_reset=1, max_steps=7, seed=13, sdiff=0.8))
    big, small, cut = mk(N), mk(64), mk(N)
    for e in (big, small, cut):
        e.reset()
        assert "k_rollout_lat<2>" in e.fragment_kernel_name()
    acts = torch.rand((T, N, 4), device="cuda")
    Ob, Rb, Tb = _bufs(T, N, 23)
    big.step_fragment(acts, Ob, Rb, Tb)
    Os, Rs, Ts = _bufs(T, 64, 23)
    small.step_fragment(acts[:, :64].contiguous(), Os, Rs, Ts)
    assert torch.equal(Ob[:, :64], Os) and torch.equal(Rb[:, :64], Rs) and torch.equal(Tb[:, :64], Ts)
    Oc, Rc, Tc = _bufs(T, N, 23)
    for lo, hi in ((0, 13), (13, 14), (14, 40)):
        cut.step_fragment(acts[lo:hi], Oc[lo:hi], Rc[lo:hi], Tc[lo:hi])
    assert torch.equal(Ob, Oc) and torch.equal(Rb, Rc) and torch.equal(Tb, Tc), "cuts: max |d obs| %.3e" % float((Ob - Oc).abs().max())
    for x, y in zip(big.get_state(), cut.get_state()):
        assert torch.equal(x, y)
    for x, y in zip(big.get_state(), small.get_state()):
        assert torch.equal(x[:64], y)
    assert int(Tb.sum()) > 0


def test_config5_fragment_is_batch_and_cut_invariant(qd):
    """the sensor pipeline (rows a round late, the extra half round at the end of every fragment) must not make results depend
    on where a run is cut: 40 steps at once == 13 + 1 + 26, and 20000 envs == the first 64 alone... except that config 5's
    waypoint phase depends on the batch size (2 pi i / N), so batch invariance is checked on a static-reference variant"""
    L, T = qd._lib, 40
    mk = lambda k: qd.dev.DeviceEnv(make_cfg(L, k, load=True, obs="LocalFrameFullStateEnv", reward="distance_energy_reward_pendulum_en4",
                                             start=1, random_params=0, auto_reset=1, max_steps=7, seed=13, sdiff=0.8))
    big, small, cut = mk(20000), mk(64), mk(20000)
    small.set_option(L.OPT_LATENCY_KERNEL, 0)   # bit-exactness across batch sizes holds within one kernel family: k_rollout_coop above 16384 envs
    for e in (big, small, cut):
        e.reset()
    assert "k_rollout_coop<2,2>" in big.fragment_kernel_name() and "k_rollout_coop<2,1>" in small.fragment_kernel_name()
    acts = torch.rand((T, 20000, 4), device="cuda")
    Ob, Rb, Tb = _bufs(T, 20000, 23)
    big.step_fragment(acts, Ob, Rb, Tb)
    Oc, Rc, Tc = _bufs(T, 20000, 23)
    for lo, hi in ((0, 13), (13, 14), (14, 40)):
        cut.step_fragment(acts[lo:hi], Oc[lo:hi], Rc[lo:hi], Tc[lo:hi])
    assert torch.equal(Tb, Tc) and torch.equal(Rb, Rc)
    assert torch.equal(Ob, Oc), "cuts: max |d obs| %.3e" % float((Ob - Oc).abs().max())
    for x, y in zip(big.get_state(), cut.get_state()):
        assert torch.equal(x, y)
    # the two register budgets are two compilations: rounding-level agreement, flags exact
    Os, Rs, Ts = _bufs(T, 64, 23)
    small.step_fragment(acts[:, :64].contiguous(), Os, Rs, Ts)
    assert torch.equal(Tb[:, :64], Ts)
    _close_rows(Ob[:, :64], Os, 12, "OCC 2 vs OCC 1")


@pytest.mark.parametrize("obs,reward,acc_at", [("BaseDroneEnv", "default_reward_fcn", 16),
                                               ("LocalFramePRYaccParamsEnv", "reward_2", 14),
                                               ("LocalFrameFullStateZvecEnv", "distance_energy_reward_pendulum_en2", 13),
                                               ("GlobalFrameRPYEnv", "reward_pendulumDistHeading", None),
                                               ("LocalFrameRmParamsEnv", "reward_3", None)])
def test_generic_fragment_equals_per_step_kernel(qd, obs, reward, acc_at):
    """k_rollout_coop<SPEC_GENERIC_FS1> / k_rollout_lat<SPEC_GENERIC_FS1>: any observation variant / reward of the load model,
    dispatched at run time in the epilogue waves, with and without the accelerometer in the row; against the per-step kernel
    k_step<true,64,4>"""
    L, n, T = qd._lib, 700, 30
    mk = lambda: qd.dev.DeviceEnv(make_cfg(L, n, load=True, obs=obs, reward=reward, start=1, random_params=1, auto_reset=1,
                                             max_steps=8, seed=17))
    a, b = mk(), mk()
    a.reset(); b.reset()
    assert "k_rollout_lat<4>" in a.fragment_kernel_name(), a.fragment_kernel_name()   # at this size, with or without the sensor in the row
    D = a.D
    g = torch.Generator(device="cuda").manual_seed(2)
    for rep in range(2):
        acts = torch.rand((T, n, 4), generator=g, device="cuda")
        O, R, Tr = _bufs(T, n, D)
        a.step_fragment(acts, O, R, Tr)
        for t in range(T):
            o, r, tr = b.step(acts[t])
            assert torch.equal(Tr[t], tr), (rep, t)
            d = (O[t] - o).abs()
            if obs != "BaseDroneEnv":
                d[:, 5] = torch.minimum(d[:, 5], (d[:, 5] - 2 * np.pi).abs())
            if acc_at is not None:
                assert float(d[:, acc_at:acc_at + 3].max()) <= 3e-3, (rep, t, float(d[:, acc_at:acc_at + 3].max()))
                d[:, acc_at:acc_at + 3] = 0
            assert float(d.max()) <= 1e-4, (rep, t, float(d.max()))
            assert float((R[t] - r).abs().max()) <= 5e-4 * max(1.0, float(r.abs().max())), (rep, t)
    assert int(Tr.sum()) > 0


def test_config5_full_size_fragment_vs_oracle_200_steps(qd, orc):
    """BASELINE config 5 at its quoted size THROUGH THE PERSISTENT KERNEL: all 8192 envs, 200 steps as four 50-step fragments against
    the float64 oracle (state_difficulty 0.8 starts, U[0,1) rotor commands, no resets): state <= 1e-4 at step 200, the
    accelerometer the rows carry, and every step's rows against the oracle's observation of its own state"""
    import bench
    n, steps, F, L = 8192, 200, 50, qd._lib
    env, _ = bench.make_env("config5", n, 42, "cuda:0", auto_reset=False)
    env.vector_reset_tensor()
    assert env._dev.fragment_kernel_name() == "qd::k_rollout_lat<2>"
    q0, v0, a0, _, _ = [x.cpu().numpy().astype(np.float64) for x in env._dev.get_state()]
    raw = env._dev.get_params().cpu().numpy()
    ob = orc.Batch(raw, True, L.OBS_KINDS.index("LocalFrameFullStateEnv"), L.REWARD_KINDS.index("distance_energy_reward_pendulum_en4"),
                   0.01, 1, 1, (0, 0, 15, 0), 1e9, 10 ** 6)
    ob.qpos[:], ob.qvel[:], ob.act[:] = q0, v0, a0
    g = torch.Generator(device="cuda").manual_seed(11)
    O, R, Tr = _bufs(F, n, 23)
    div = Divergence(True)
    worst_acc = 0.0
    for f in range(steps // F):
        acts = torch.rand((F, n, 4), generator=g, device="cuda")
        env.step_fragment_tensor(acts, O, R, Tr)
        ah, Oh = acts.cpu().numpy().astype(np.float64), O.cpu().numpy().astype(np.float64)
        for t in range(F):
            ob.step(ah[t], threads=8)
            # the row's sensor entries are the reading mj_step computed in this step (quirk C-6), which the oracle keeps too
            worst_acc = max(worst_acc, float(np.max(np.abs(Oh[t][:, 12:15] - ob.sensor) / np.maximum(1.0, np.abs(ob.sensor)))))
        gq, gv, ga, gs, _ = [x.cpu().numpy().astype(np.float64) for x in env._dev.get_state()]
        div.update(dict(qpos=gq, qvel=gv, act=ga), dict(qpos=ob.qpos, qvel=ob.qvel, act=ob.act))
    print(div.table("config 5 through k_rollout_coop<SPEC_LSTM>, 8192 envs, 200 steps vs the float64 oracle"))
    print("accelerometer entries of the rows vs the oracle's sensor, max relative over 200 steps x 8192 envs: %.3e" % worst_acc)
    assert div.max("mixed") < 1e-4 and div.max("rel") < 1e-4
    assert worst_acc < 2e-3
    np.testing.assert_allclose(gs, ob.sensor, rtol=2e-4, atol=2e-3)


# ------------------------------------------------------------------ the single-body model: fragments run in k_rollout
def test_config2_full_size_fragment_vs_oracle_and_per_step(qd, orc):
    """BASELINE config 2 at its full size through qd_step_fragment (k_rollout_pair: a physics wavefront and an epilogue wavefront per
    64 envs, the state in registers for the whole fragment): 4096 SimpleDrone envs, fixed initial state, 200 steps of U[0.5, 1) rotor
    actions as four 50-step fragments; every env against the float64 oracle (<= 1e-4 relative), every step's rows, rewards and
    flags against the oracle's and against the per-step kernel's"""
    rng = np.random.default_rng(321)
    n, L, steps, F = 4096, qd._lib, 200, 50
    c = make_cfg(L, n, load=False, obs="SimpleDrone", reward="simple_drone_reward", frame_skip=2, h=0.001, ctrl_map=0, term=1,
                 ref=(0, 0, 1, 0), start_pos=(0, 0, 1, 0), max_steps=10 ** 6, max_distance=1e9)
    env, per = qd.dev.DeviceEnv(c), qd.dev.DeviceEnv(c)
    assert env.fragment_kernel_name() == "qd::k_rollout_pair", env.fragment_kernel_name()   # (<= 16384 envs; k_rollout<false,64,3> above)
    raw = np.tile([1.35, 0.15, 7.5, 0.015, 0, 0], (n, 1))
    q0 = np.tile([0, 0, 1, 1, 0, 0, 0.0], (n, 1))
    for e in (env, per):
        e.set_params(raw)
        e.set_state(q0, np.zeros((n, 6)), np.zeros((n, 4)))
    ob = orc.Batch(raw, False, L.OBS_KINDS.index("SimpleDrone"), L.REWARD_KINDS.index("simple_drone_reward"), 0.001, 2, 0, (0, 0, 1, 0), 1e9, 10 ** 6)
    ob.qpos[:], ob.qvel[:], ob.act[:] = q0, 0.0, 0.0
    acts = rng.uniform(0.5, 1.0, (steps, n, 4)).astype(np.float32)
    dacts = torch.as_tensor(acts).cuda()
    O, R, Tr = _bufs(F, n, 6)
    wo = wr = wp = 0.0
    for f in range(steps // F):
        env.step_fragment(dacts[f * F:(f + 1) * F].contiguous(), O, R, Tr)
        Oh, Rh, Th = O.cpu().numpy().astype(np.float64), R.cpu().numpy().astype(np.float64), Tr.cpu().numpy()
        for t in range(F):
            oo, orr, otr = ob.step(acts[f * F + t].astype(np.float64), threads=8)
            o, r, tr = per.step(dacts[f * F + t])
            d = np.abs(Oh[t] - oo)
            d[:, 3:] = np.minimum(d[:, 3:], np.abs(d[:, 3:] - 2 * np.pi))      # the three angles wrap at +-pi
            dp = (O[t] - o).abs()
            dp[:, 3:] = torch.minimum(dp[:, 3:], (dp[:, 3:] - 2 * np.pi).abs())
            wo, wr = max(wo, float(d.max())), max(wr, float(np.abs(Rh[t] - orr).max()))
            wp = max(wp, float(dp.max()), float((R[t] - r).abs().max()))
            assert np.array_equal(Th[t].astype(bool), otr.astype(bool)) and torch.equal(Tr[t], tr)
    gq, gv, ga, _, gk = [x.cpu().numpy().astype(np.float64) for x in env.get_state()]
    assert np.all(gk == steps)
    err = max(float(np.max(np.abs(g - w) / np.maximum(1.0, np.abs(w)))) for g, w in ((gq, ob.qpos), (gv, ob.qvel), (ga, ob.act)))
    print("config 2 through k_rollout, 4096 envs, 200 steps: max relative state divergence %.3e; rows |obs - oracle| %.3e, "
          "|reward - oracle| %.3e; against the per-step kernel %.3e" % (err, wo, wr, wp))
    assert err < 1e-4 and wo < 2e-4 and wr < 2e-4 and wp < 2e-5


@pytest.mark.parametrize("n", [300, 20000])
def test_simple_drone_fragments_with_resets(qd, n):
    """SimpleDrone.reset_model sampling inside a fragment (QD_START_SIMPLE: qpos0 + U(-0.03, 0.03), only drone 0 moved to start_pos)
    and the BaseDroneEnv observation variants on the single-body model: fragments equal per-step launches -- the two-wavefront
    kernel (<= 16384 envs) and the single-wavefront one above"""
    L, T = qd._lib, 40
    cases = [dict(load=False, obs="SimpleDrone", reward="simple_drone_reward", frame_skip=2, h=0.001, ctrl_map=0, term=1,
                  ref=(0, 0, 1, 0), start_pos=(0, 0, 1, 0), start=2, auto_reset=1, max_steps=9, max_distance=4.0),
             dict(load=False, obs="LocalFrameRPYEnv", reward="distance_energy_reward", start=1, random_params=1, auto_reset=1, max_steps=9)]
    for kw in cases:
        mk = lambda: qd.dev.DeviceEnv(make_cfg(L, n, seed=4, **kw))
        a, b = mk(), mk()
        a.reset(); b.reset()
        assert a.fragment_kernel_name() == ("qd::k_rollout_pair" if (kw["obs"] == "SimpleDrone" and n <= 16384) else
                                            "qd::k_rollout<false,64,%d>" % (3 if kw["obs"] == "SimpleDrone" else 4)), a.fragment_kernel_name()
        lo = 0.5 if kw["obs"] == "SimpleDrone" else 0.0
        for rep in range(2):
            acts = lo + (1 - lo) * torch.rand((T, n, 4), device="cuda")
            O, R, Tr = _bufs(T, n, a.D)
            a.step_fragment(acts, O, R, Tr)
            for t in range(T):
                o, r, tr = b.step(acts[t])
                assert torch.equal(Tr[t], tr), (kw["obs"], rep, t)
                d = (O[t] - o).abs()
                d[:, 3:6] = torch.minimum(d[:, 3:6], (d[:, 3:6] - 2 * np.pi).abs())
                assert float(d.max()) <= 2e-5 and float((R[t] - r).abs().max()) <= 2e-5, (kw["obs"], rep, t, float(d.max()))
        assert int(Tr.sum()) > 0


def test_generic_and_pid_instantiations_at_two_workgroups_per_cu(qd):
    """the register-capped (OCC = 2) instantiations of the run-time-dispatched kernel and of the PID-driven sensor-carrying kernel,
    which only run above 16384 envs: against the per-step launches on the first envs of a 20 000-env batch"""
    L, n, T = qd._lib, 20000, 24
    mk = lambda: qd.dev.DeviceEnv(make_cfg(L, n, load=True, obs="LocalFramePRYaccEnv", reward="reward_1", start=1, random_params=1,
                                             auto_reset=1, max_steps=8, seed=23))
    a, b = mk(), mk()
    a.reset(); b.reset()
    assert a.fragment_kernel_name() == "qd::k_rollout_coop<4,2>"
    acts = torch.rand((T, n, 4), device="cuda")
    O, R, Tr = _bufs(T, n, a.D)
    a.step_fragment(acts, O, R, Tr)
    for t in range(T):
        o, r, tr = b.step(acts[t])
        assert torch.equal(Tr[t], tr), t
        d = _heading_safe_absdiff(O[t], o)
        assert float(d[:, 12:15].max()) <= 3e-3, (t, float(d[:, 12:15].max()))
        d[:, 12:15] = 0
        assert float(d.max()) <= 1e-4 and float((R[t] - r).abs().max()) <= 1e-3, (t, float(d.max()))
    # the PID cascade on train_LSTM.py's configuration (sensor-carrying rows, circle waypoints), persistent against launch by launch
    import bench
    e1, _ = bench.make_env("config5", n, 42, "cuda:0")
    e2, _ = bench.make_env("config5", n, 42, "cuda:0")
    e2._dev.set_option(L.OPT_PERSISTENT_FRAGMENTS, 0)
    e1.vector_reset_tensor(); e2.vector_reset_tensor()
    e1.pid_reset(); e2.pid_reset()
    o1, r1, t1, a1 = e1.rollout_pid_tensor(40, want_actions=True)
    o2, r2, t2, a2 = e2.rollout_pid_tensor(40, want_actions=True)
    assert torch.equal(t1, t2)
    assert float((a1 - a2).abs().max()) <= 2e-4 and float((r1 - r2).abs().max()) <= 2e-3
    d = _heading_safe_absdiff(o1, o2)
    assert float(d[..., 12:15].max()) <= 5e-3
    d[..., 12:15] = 0
    assert float(d.max()) <= 3e-4, float(d.max())


def test_pid_fragment_rows_that_carry_the_activations_and_fragment_cuts(qd):
    """The PID-driven persistent kernel applies the controller's action a round late (the motors are filters: the step does not wait
    for it).  Raw observation rows (BaseDroneEnv._get_obs) carry the activations, so they show whether every wave sees the same
    filtered values: against T x (qd_pid_action, qd_step), with resets every 7 steps, and the same rollout cut into three
    fragments (the last filter step is applied when a fragment ends) bit for bit."""
    L, n, T = qd._lib, 1000, 36
    mk = lambda: qd.dev.DeviceEnv(make_cfg(L, n, load=True, obs="BaseDroneEnv", reward="default_reward_fcn", start=1, random_params=1,
                                             auto_reset=1, max_steps=7, seed=31))
    a, b = mk(), mk()
    for e in (a, b):
        e.reset(); e.pid_reset()
    assert a.fragment_kernel_name() == "qd::k_rollout_lat<4>"   # (of action fragments; the PID loop below runs k_rollout_coop<4,1,true>)
    O, R, Tr, A = a.rollout_pid(T, want_actions=True)
    worst = 0.0
    for t in range(T):
        act = b.pid_action()
        o, r, tr = b.step(act)
        b.pid_reset(tr)          # a new episode starts with fresh controller objects (the rollout kernels do this in the step)
        assert torch.equal(Tr[t], tr), t
        np.testing.assert_allclose(A[t].cpu().numpy(), act.cpu().numpy(), atol=2e-4, err_msg="t=%d" % t)
        d = _heading_safe_absdiff(O[t], o)
        worst = max(worst, float(d.max()))
        assert float(d.max()) <= 3e-3 and float((R[t] - r).abs().max()) <= 2e-3, (t, float(d.max()))
    assert int(Tr.sum()) == (T // 7) * n
    for cuts in ((5, 19, 12), (7, 14, 1, 14)):          # the second: fragments that END on a reset step (7, 21) and a one-step fragment
        c = mk()
        c.reset(); c.pid_reset()
        parts = [c.rollout_pid(k, want_actions=True) for k in cuts]
        for j, name in enumerate(("obs", "reward", "truncated", "actions")):
            got = torch.cat([p[j] for p in parts])
            assert torch.equal(got, (O, R, Tr, A)[j]), (cuts, name)
        for x, y in zip(a.get_state(), c.get_state()):
            assert torch.equal(x, y), cuts
