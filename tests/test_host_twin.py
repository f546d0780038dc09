"""The product's templated device headers (csrc/qd_{model,dynamics,obsrew}.h) instantiated on the HOST by a
test-only harness (tests/host_twin), so that the specialised body-frame derivation the HIP kernels run can be
compared with the general world-frame oracle at 1e-9 here, without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CENTER = np.array([1, 0.17, 7, 0.01, 1.2, 0.3])
WIDTH = np.array([0.1, 0.02, 1, 0.0025, 0.2, 0.05])
dp = C.POINTER(C.c_double)


def P(a):
    return a.ctypes.data_as(dp)


@pytest.fixture(scope="module")
def twin():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "qd_host_twin.so")
    src = os.path.join(ROOT, "tests", "host_twin", "qd_host_twin.cpp")
    inc = os.path.join(ROOT, "mujoco-drone_amd", "csrc")
    deps = [src] + [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", inc, "-o", so, src])
    lib = C.CDLL(so)
    lib.twin_reward.restype = C.c_double
    lib.twin_reward_q.restype = C.c_double
    lib.twin_round5.restype = C.c_double
    lib.twin_round5.argtypes = [C.c_double]
    return lib


def rand_raw(rng, load):
    r = CENTER + rng.uniform(-1, 1, 6) * WIDTH
    if not load:
        r[4:] = 0
    return r


def test_round5_matches_printf_round_trip(twin, orc):
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-3, 3, 20000), 10.0 ** rng.uniform(-6, 4, 20000), [0.0, 1.0, 0.1, 0.015, 2e-5, 0.785398163]])
    for x in xs:
        assert twin.twin_round5(float(x)) == orc.round5g(float(x)), x


@pytest.mark.parametrize("load", [1, 0])
def test_closed_form_model_vs_general_geom_composition(twin, orc, load):
    rng = np.random.default_rng(1)
    for _ in range(300):
        raw = rand_raw(rng, load)
        m = orc.build_model(raw)
        m28 = np.zeros(28)
        assert twin.twin_derive(P(raw), P(m28)) == load
        want = [m.m0, m.c0[2], m.I0full[0], m.I0full[1], m.I0full[2], m.rotor[1][0], m.gearF, m.gearT[0], 1 / m.tau]
        np.testing.assert_allclose(m28[:9], want, rtol=1e-12)
        np.testing.assert_allclose(m28[9:13], [m.m2, m.lc, m.I2[0], m.I2[2]], rtol=1e-12, atol=1e-300)
        # symmetry facts the closed form relies on
        assert max(abs(m.I0full[3]), abs(m.I0full[4]), abs(m.I0full[5])) < 1e-12
        assert abs(m.c0[0]) < 1e-15 and abs(m.c0[1]) < 1e-15
        rot = np.array([[m.rotor[i][j] for j in range(3)] for i in range(4)])
        np.testing.assert_array_equal(np.abs(rot[:, :2]), np.full((4, 2), m.rotor[1][0]))
        assert list(np.sign(rot[:, 0])) == [1, 1, -1, -1] and list(np.sign(rot[:, 1])) == [-1, 1, 1, -1]
        assert list(np.sign(m.gearT[:])) == [1, -1, 1, -1]
        # MuJoCo's principal frame of the core is a signed axis permutation (so body-axis drag is exact)
        R0 = np.array(m.R0i[:]).reshape(3, 3)
        assert np.allclose(np.sort(np.abs(R0).ravel()), [0] * 6 + [1] * 3, atol=1e-9)
        # fluid coefficients against the oracle's box dimensions, permuted back to body axes
        perm = np.argmax(np.abs(R0), axis=0)          # inertial axis k lies along body axis perm[k]
        box = np.zeros(3); box[perm] = np.array(m.box0[:])
        d = box.mean()
        np.testing.assert_allclose(m28[13:15], [3 * np.pi * d * 2e-5, np.pi * d ** 3 * 2e-5], rtol=1e-10)
        np.testing.assert_allclose(m28[15:18], [0.6 * box[1] * box[2], 0.6 * box[0] * box[2], 0.6 * box[0] * box[1]], rtol=1e-10)
        np.testing.assert_allclose(m28[18:21], [1.2 * box[i] * (box[(i + 1) % 3] ** 4 + box[(i + 2) % 3] ** 4) / 64 for i in range(3)], rtol=1e-10)


@pytest.mark.parametrize("load", [1, 0])
def test_specialised_dynamics_vs_general_oracle_f64(twin, orc, load):
    rng = np.random.default_rng(2)
    worst = 0.0
    for _ in range(200):
        raw = rand_raw(rng, load)
        m = orc.build_model(raw)
        m28 = np.zeros(28); twin.twin_derive(P(raw), P(m28))
        nq, nv = (9, 8) if load else (7, 6)
        qpos = np.zeros(nq); qpos[:3] = [0, 0, 15] + rng.normal(size=3)
        q = rng.normal(size=4); qpos[3:7] = q / np.linalg.norm(q)
        if load:
            qpos[7:] = rng.normal(scale=0.6, size=2)
        qvel = rng.normal(scale=1.5, size=nv)
        act, ctrl = rng.uniform(0, 1, 4), rng.uniform(-0.1, 1.1, 4)
        want = orc.step(m, 0.01, 3, qpos, qvel, act, ctrl)
        a, b, c, s = qpos.copy(), qvel.copy(), act.copy(), np.zeros(3)
        twin.twin_step_f64(load, P(m28), P(a), P(b), P(c), P(ctrl), C.c_double(0.01), 3, P(s))
        worst = max(worst, *(np.abs(g - w).max() for g, w in zip((a, b, c, s), want)))
    assert worst < 1e-9, worst


def test_float32_device_arithmetic_200_step_divergence(twin, orc):
    """what the GPU computes, predicted on the CPU: float32 state / trig / drag / integration with the
    float64 algebra core, 200 steps of random rotor commands, against the float64 oracle.  Target <= 1e-4."""
    rng = np.random.default_rng(5)
    worst = 0.0
    for i in range(24):
        raw = rand_raw(rng, 1)
        m = orc.build_model(raw)
        m28 = np.zeros(28); twin.twin_derive(P(raw), P(m28))
        qpos = np.zeros(9); qpos[:3] = [0, 0, 15] + rng.normal(size=3); qpos[3] = 1; qpos[7:] = rng.normal(scale=0.2, size=2)
        qvel = rng.normal(scale=0.4, size=8)
        f32 = lambda x: x.astype(np.float32).astype(np.float64)
        a, b, c, s = f32(qpos), f32(qvel), np.zeros(4), np.zeros(3)
        oq, ov, oa = a.copy(), b.copy(), c.copy()
        for t in range(200):
            ctrl = f32(0.1 + 0.9 * rng.uniform(0, 1, 4))
            oq, ov, oa, _ = orc.step(m, 0.01, 1, oq, ov, oa, ctrl)
            twin.twin_step_f32(1, P(m28), P(a), P(b), P(c), P(ctrl), C.c_double(0.01), 1, P(s))
            a, b, c = f32(a), f32(b), f32(c)
        for g, w in ((a, oq), (b, ov), (c, oa)):
            worst = max(worst, float(np.max(np.abs(g - w) / np.maximum(1, np.abs(w)))))
    assert worst < 1e-4, worst


def test_observation_reward_headers_vs_golden(twin, golden, orc):
    ref, A, K = golden["st_ref"], golden["st_actions"], golden["st_num_steps"]
    for tag in ("33", "29"):
        S = golden["st" + tag]
        for kind in range(15):
            name = orc.OBS_KINDS[kind]
            if kind == 12:
                continue
            key = "obs%s_%s" % (tag, name)
            for i in range(len(S)):
                out = np.zeros(40)
                n = twin.twin_obs(kind, P(np.ascontiguousarray(S[i])), int(tag), P(ref), P(out))
                want = golden[key][i] if kind > 0 else S[i]
                assert n == len(want) == twin.twin_obs_dim(kind, int(tag))
                np.testing.assert_allclose(out[:n], want, atol=1e-11)
    S = golden["st33"]
    for kind, name in enumerate(orc.REWARD_KINDS[:17]):
        got = [twin.twin_reward(kind, P(np.ascontiguousarray(S[i])), P(np.ascontiguousarray(A[i])), int(K[i]), P(ref),
                                C.c_double(4.0)) for i in range(len(S))]
        np.testing.assert_allclose(got, golden["rew_" + name], rtol=1e-10, atol=1e-10)
    tr = [bool(twin.twin_truncated(P(np.ascontiguousarray(S[i])), P(ref), int(K[i]), C.c_double(4.0), 512)) for i in range(len(S))]
    assert tr == list(golden["trunc33"])


def test_quaternion_matrix_fast_path_equals_rpy_path(twin, orc):
    """fused kernels take the attitude matrix from the quaternion instead of rebuilding it from roll/pitch/yaw"""
    rng = np.random.default_rng(9)
    ref = np.array([0.3, -0.2, 15.0, 0.7])
    for _ in range(50):
        qpos = np.zeros(9); qpos[:3] = [0, 0, 15] + rng.normal(size=3)
        q = rng.normal(size=4); qpos[3:7] = q / np.linalg.norm(q); qpos[7:] = rng.normal(scale=0.5, size=2)
        qvel, sens, act, par = rng.normal(size=8), rng.normal(size=3), rng.uniform(0, 1, 4), CENTER.copy()
        s = orc.drone_state(1, qpos, qvel, sens, act, ref, par)
        a = rng.uniform(0, 1, 4)
        for kind in range(1, 15):
            if kind == 12:
                continue
            out = np.zeros(40)
            n = twin.twin_obs_q(kind, P(qpos), P(qvel), P(sens), P(act), P(ref), P(par), P(out))
            np.testing.assert_allclose(out[:n], orc.obs(kind, s, ref), atol=1e-9, err_msg=str(kind))
        for kind in range(17):
            got = twin.twin_reward_q(kind, P(qpos), P(qvel), P(sens), P(act), P(ref), P(par), P(a), 37, C.c_double(4.0))
            assert abs(got - orc.reward(kind, s, a, 37, ref, 4.0)) < 1e-9, kind


@pytest.mark.parametrize("prec,tol", [("f64", 1e-11), ("f32", 2e-5)])
def test_pid_header_vs_reference_controllers(twin, golden, prec, tol):
    """csrc/qd_pid.h (what the GPU runs) against the reference's PositionController / AttittudeController outputs."""
    G = golden
    fn = getattr(twin, "twin_pid_" + prec)
    nd, T = len(G["pid_masses"]), G["pid_xyz"].shape[1]
    ref = np.ascontiguousarray(G["pid_ref"])
    for d in range(nd):
        st = np.zeros(13); st[12] = 3
        for t in range(T):
            xyz = np.ascontiguousarray(G["pid_xyz"][:, t, d]); rpy = np.ascontiguousarray(G["pid_rpy"][:, t, d])
            pa, rz, ct, ac = np.zeros(3), np.zeros(4), np.zeros(4), np.zeros(4)
            fn(P(st), P(ref), P(xyz), P(rpy), C.c_double(G["pid_masses"][d]), C.c_double(G["pid_forces"][d]),
               P(pa), P(rz), P(ct), P(ac))
            np.testing.assert_allclose(pa, G["pid_pos_action"][t][:, d], rtol=0, atol=tol * 50)  # D gain * 50 Hz
            np.testing.assert_allclose(rz, G["pid_rpyz"][t][:, d], rtol=0, atol=tol * 50)
            np.testing.assert_allclose(ct, G["pid_ctrl"][t][d], rtol=0, atol=tol * 50)
            np.testing.assert_allclose(ac, np.clip(G["pid_ctrl"][t][d] - 0.1, 0, 1), rtol=0, atol=tol * 50)
        assert st[12] == 0


# ------------------------------------------------------------------ floor contact, single-body model
def _rand_pose_near_floor(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    if rng.uniform() < 0.5:                                  # nearly level, as when landing
        ang = rng.normal(scale=0.15, size=3)
        q = np.array([1.0, *(0.5 * ang)]); q /= np.linalg.norm(q)
    return np.array([rng.normal(scale=0.3), rng.normal(scale=0.3), rng.uniform(-0.01, 0.12), *q])


def test_floor_contact_generation_twin_vs_oracle(twin, orc):
    """the product's contact generation (qd_contact.h) and the oracle's list the same contact points for random poses near the
    floor: boxes, upright and tilted cylinders, any attitude"""
    rng = np.random.default_rng(21)
    twin.twin_floor_contacts.restype = C.c_int
    twin.twin_floor_contacts.argtypes = [C.c_double, dp, dp]
    seen = 0
    for _ in range(400):
        raw = rand_raw(rng, False)
        m = orc.build_model(raw)
        qpos = _rand_pose_near_floor(rng)
        want = orc.floor_contacts(m, qpos)
        out = np.zeros(4 * 64)
        n = twin.twin_floor_contacts(float(raw[1]), P(qpos), P(out))
        assert n == len(want)
        got = out[:4 * n].reshape(n, 4)
        for g, (pos, dist, body) in zip(got, want):
            np.testing.assert_allclose(g[:3], pos, atol=1e-12)
            assert abs(g[3] - dist) < 1e-12 and body == 0
        seen += n
    assert seen > 1000


def test_floor_contact_solve_twin_vs_oracle(twin, orc):
    """two routes to the minimiser of the same convex contact problem: the product's Newton method on the primal (6 unknowns,
    COM coordinates, diagonal inertia) against the oracle's projected Gauss-Seidel on the dual (general Jacobians, full mass
    matrix): accelerations and normal force agree for random poses / velocities in contact"""
    rng = np.random.default_rng(22)
    twin.twin_forward_floor.restype = C.c_int
    twin.twin_forward_floor.argtypes = [dp, C.c_double, dp, dp, dp, C.c_double, dp, dp]
    checked = 0
    worst = 0.0
    for _ in range(300):
        raw = rand_raw(rng, False)
        m = orc.build_model(raw)
        m16 = np.zeros(32)
        twin.twin_derive(P(np.asarray(raw, dtype=np.float64)), P(m16))
        qpos = _rand_pose_near_floor(rng)
        qvel = np.concatenate([rng.normal(scale=0.5, size=3), rng.normal(scale=1.0, size=3)])
        act = rng.uniform(0, 1, 4)
        h = float(rng.choice([0.01, 0.001]))
        want, n_o, fz_o = orc.forward_floor(m, qpos, qvel, act, h)
        got, fz = np.zeros(6), np.zeros(1)
        n = twin.twin_forward_floor(P(m16), float(raw[1]), P(qpos), P(qvel), P(act), h, P(got), P(fz))
        assert n == n_o
        if fz_o > 0:
            checked += 1
            scale = max(1.0, float(np.abs(want).max()))
            worst = max(worst, float(np.abs(got - want).max()) / scale)
            np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5 * scale)
            assert abs(fz[0] - fz_o) < 2e-5 * max(1.0, fz_o)
        else:
            np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)
    assert checked > 100
    print("floor contact: %d states in contact, worst relative difference %.2e" % (checked, worst))


def test_floor_contact_tree_twin_vs_oracle(twin, orc):
    """the load model: drone + tether + load box near the floor (the box, the rod or the airframe touching), the product's 8-unknown
    Newton method with its own mass-matrix assembly against the oracle's dual solver: constrained accelerations and normal force"""
    rng = np.random.default_rng(23)
    twin.twin_forward_floor_tree.restype = C.c_int
    twin.twin_forward_floor_tree.argtypes = [dp, dp, dp, dp, dp, C.c_double, dp, dp, dp]
    checked, worst = 0, 0.0
    for trial in range(300):
        raw = np.asarray(rand_raw(rng, True), dtype=np.float64)
        m = orc.build_model(raw)
        m16 = np.zeros(32)
        twin.twin_derive(P(raw), P(m16))
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        if trial % 2:
            ang = rng.normal(scale=0.2, size=3); q = np.array([1.0, *(0.5 * ang)]); q /= np.linalg.norm(q)
        z = rng.uniform(0.0, 0.3) if trial % 3 == 0 else raw[4] + rng.uniform(-0.15, 0.25)     # airframe low, or the load near the floor
        qpos = np.array([rng.normal(scale=0.3), rng.normal(scale=0.3), z, *q, rng.normal(scale=0.4), rng.normal(scale=0.4)])
        qvel = np.concatenate([rng.normal(scale=0.5, size=3), rng.normal(scale=1.0, size=3), rng.normal(scale=1.0, size=2)])
        act = rng.uniform(0, 1, 4)
        h = float(rng.choice([0.01, 0.002]))
        want, n_o, fz_o = orc.forward_floor(m, qpos, qvel, act, h)
        got, gimp, fz = np.zeros(8), np.zeros(8), np.zeros(1)
        n = twin.twin_forward_floor_tree(P(m16), P(raw), P(qpos), P(qvel), P(act), h, P(got), P(gimp), P(fz))
        assert n == n_o, (trial, n, n_o)
        scale = max(1.0, float(np.abs(want).max()))
        if fz_o > 0:
            checked += 1
            worst = max(worst, float(np.abs(got - want).max()) / scale)
            np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6 * scale)
            assert abs(fz[0] - fz_o) < 1e-6 * max(1.0, fz_o)
            # the Euler step's accelerations: (M + h D) qimp = M qacc, checked through the oracle's mass matrix
            Mm = orc.mass_matrix(m, qpos)
            Mh = Mm.copy(); Mh[6, 6] += h * 0.15; Mh[7, 7] += h * 0.15
            np.testing.assert_allclose(gimp, np.linalg.solve(Mh, Mm @ want), rtol=1e-6, atol=1e-6 * scale)
        else:
            np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-8 * scale)
    assert checked > 100
    print("floor contact (load model): %d states in contact, worst relative difference %.2e" % (checked, worst))


def test_three_assemblies_of_the_mass_matrix_agree(twin, orc):
    """the load model's 8 x 8 mass matrix exists three times, written independently: the oracle's projected Newton-Euler assembly,
    the product's contact path (COM Jacobians of the three bodies, qd_contact.h) and -- implicitly, never formed -- the body-frame
    elimination of qd_dynamics.h (checked through accelerations elsewhere).  The two explicit ones must agree to rounding"""
    rng = np.random.default_rng(24)
    twin.twin_tree_mass_matrix.argtypes = [dp, dp, dp]
    for _ in range(50):
        raw = np.asarray(rand_raw(rng, True), dtype=np.float64)
        m = orc.build_model(raw)
        m16 = np.zeros(32)
        twin.twin_derive(P(raw), P(m16))
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        qpos = np.array([*rng.normal(size=3), *q, rng.normal(scale=0.7), rng.normal(scale=0.7)])
        got = np.zeros(64)
        twin.twin_tree_mass_matrix(P(m16), P(qpos), P(got))
        want = orc.mass_matrix(m, qpos)
        np.testing.assert_allclose(got.reshape(8, 8), want, rtol=1e-9, atol=1e-11)


def test_cooperative_pieces_equal_the_monolithic_forward(twin):
    """qd_dynamics.h holds the load model's forward dynamics twice: forward(), which the single-lane kernels run, and the
    pieces (applied / inertial wrench, mass factor, reduce, finish) k_step_coop runs in three wavefronts.  forward_pieces()
    composes the pieces in one lane; here both are evaluated on the host for random states: the same accelerations and
    accelerometer reading to 1e-11 in float64 (same algebra, different association), and in float32 to the rounding level the
    GPU parity tests allow per step."""
    rng = np.random.default_rng(77)
    worst64, worst32 = 0.0, 0.0
    for k in range(400):
        raw = rand_raw(rng, 1)
        m28 = np.zeros(28)
        assert twin.twin_derive(P(raw), P(m28)) == 1
        qpos = np.zeros(9); qpos[:3] = rng.uniform(-2, 2, 3) + [0, 0, 15]
        q = rng.normal(size=4); qpos[3:7] = q / np.linalg.norm(q) * rng.uniform(0.9, 1.1)   # unnormalised on purpose
        qpos[7:] = rng.normal(0, 0.6, 2)
        qvel = np.concatenate([rng.normal(0, 3, 3), rng.normal(0, 4, 3), rng.normal(0, 3, 2)])
        act = rng.uniform(-0.1, 1.2, 4)
        o64, o32 = np.zeros(38), np.zeros(38)
        twin.twin_forward_pair_f64(P(m28), P(qpos), P(qvel), P(act), C.c_double(0.01), P(o64))
        twin.twin_forward_pair_f32(P(m28), P(qpos), P(qvel), P(act), C.c_double(0.01), P(o32))
        scale = np.maximum(1.0, np.abs(o64[:19]))
        worst64 = max(worst64, float(np.max(np.abs(o64[:19] - o64[19:]) / scale)))
        worst32 = max(worst32, float(np.max(np.abs(o32[:19] - o32[19:]) / scale)))
    print("pieces vs monolithic forward: float64 %.2e, float32 %.2e (relative to max(1, |value|))" % (worst64, worst32))
    assert worst64 < 1e-11
    assert worst32 < 5e-5


def test_latency_pieces_equal_the_monolithic_forward(twin):
    """qd_dynamics.h's third arrangement of the load model's forward dynamics, the one k_rollout_lat runs: inertia assembled from
    per-env coefficients of the tether direction (lat_consts), hinges eliminated first, the 3x3 inverted by its adjugate and kept
    as the symmetric inverse of the 5x5 rotational + hinge system (mass_inverse / solve_inv5), the applied wrench in two halves
    (applied_core_link + applied_tether).  Same equations: the damping-implicit accelerations equal forward()'s to 1e-11 in
    float64; with float32 state / trigonometry / drag and float64 algebra (what the device runs) both are compared with the
    float64 value."""
    rng = np.random.default_rng(79)
    worst64, worst32, mono32, worst_ex = 0.0, 0.0, 0.0, 0.0
    for k in range(600):
        raw = rand_raw(rng, 1)
        m28 = np.zeros(28)
        assert twin.twin_derive(P(raw), P(m28)) == 1
        qpos = np.zeros(9); qpos[:3] = rng.uniform(-2, 2, 3) + [0, 0, 15]
        q = rng.normal(size=4); qpos[3:7] = q / np.linalg.norm(q) * rng.uniform(0.9, 1.1)
        qpos[7:] = rng.normal(0, 0.8, 2)
        qvel = np.concatenate([rng.normal(0, 3, 3), rng.normal(0, 4, 3), rng.normal(0, 3, 2)])
        act = rng.uniform(-0.1, 1.2, 4)
        o64, o32, l64, l32 = np.zeros(38), np.zeros(38), np.zeros(19), np.zeros(19)
        twin.twin_forward_pair_f64(P(m28), P(qpos), P(qvel), P(act), C.c_double(0.01), P(o64))
        twin.twin_forward_pair_f32(P(m28), P(qpos), P(qvel), P(act), C.c_double(0.01), P(o32))
        twin.twin_forward_lat_f64(P(m28), P(qpos), P(qvel), P(act), C.c_double(0.01), P(l64))
        twin.twin_forward_lat_f32(P(m28), P(qpos), P(qvel), P(act), C.c_double(0.01), P(l32))
        scale = np.maximum(1.0, np.abs(o64[8:16]))
        worst64 = max(worst64, float(np.max(np.abs(o64[8:16] - l64[:8]) / scale)))
        worst32 = max(worst32, float(np.max(np.abs(l32[:8] - o64[8:16]) / scale)))
        # the damping-explicit accelerations and the accelerometer, by the 2 x 2 correction of the implicit solve (explicit_from_implicit)
        ex_ref = np.concatenate([o64[0:8], o64[16:19]])
        worst_ex = max(worst_ex, float(np.max(np.abs(ex_ref - l64[8:19]) / np.maximum(1.0, np.abs(ex_ref)))))
        mono32 = max(mono32, float(np.max(np.abs(o32[8:16] - o64[8:16]) / scale)))
    print("latency pieces vs forward() in float64: %.2e; in float32 against the float64 value: %.2e (forward() in float32: %.2e)" % (worst64, worst32, mono32))
    assert worst64 < 1e-11
    print("explicit accelerations + accelerometer from the implicit solve vs forward() in float64: %.2e" % worst_ex)
    assert worst_ex < 1e-10
    # float32 inputs: both arrangements are judged against the float64 accelerations.  The latency pieces put the float32 sine /
    # cosine pairs back on the unit circle in float64 (trig_unit) and come out closer than forward(), whose J and B D^-1 B^T see
    # the pairs as they are (measured 7.7e-5 against 1.7e-4 over 2000 states)
    assert worst32 < 2e-4 and worst32 <= 1.5 * mono32


def test_sensor_is_affine_in_the_activations(twin):
    """sensor_affine (qd_dynamics.h): c0 + sum a_i col_i equals the accelerometer of forward() at the same state and
    activations -- what lets the reset pool prepare a new episode's first sensor reading before the activations are known"""
    rng = np.random.default_rng(78)
    worst = worst_lat = worst_lat32 = 0.0
    for k in range(300):
        raw = rand_raw(rng, 1)
        m28 = np.zeros(28)
        assert twin.twin_derive(P(raw), P(m28)) == 1
        qpos = np.zeros(9); qpos[:3] = rng.uniform(-2, 2, 3) + [0, 0, 15]
        q = rng.normal(size=4); qpos[3:7] = q / np.linalg.norm(q)
        qpos[7:] = rng.normal(0, 0.6, 2)
        qvel = np.concatenate([rng.normal(0, 3, 3), rng.normal(0, 4, 3), rng.normal(0, 3, 2)])
        act = rng.uniform(-0.2, 1.3, 4)
        o = np.zeros(6)
        twin.twin_sensor_affine_f64(P(m28), P(qpos), P(qvel), P(act), C.c_double(0.01), P(o))
        worst = max(worst, float(np.max(np.abs(o[:3] - o[3:]) / np.maximum(1.0, np.abs(o[3:])))))
        # the latency arrangement's form (the closed policy loop's reset lanes): float64 against forward(), float32 against float64
        l64, l32 = np.zeros(3), np.zeros(3)
        twin.twin_sensor_affine_lat_f64(P(m28), P(qpos), P(qvel), P(act), C.c_double(0.01), P(l64))
        twin.twin_sensor_affine_lat_f32(P(m28), P(qpos), P(qvel), P(act), C.c_double(0.01), P(l32))
        worst_lat = max(worst_lat, float(np.max(np.abs(l64 - o[3:]) / np.maximum(1.0, np.abs(o[3:])))))
        worst_lat32 = max(worst_lat32, float(np.max(np.abs(l32 - o[3:]) / np.maximum(1.0, np.abs(o[3:])))))
    print("affine sensor vs forward(): %.2e; latency form %.2e (float64), %.2e (float32)" % (worst, worst_lat, worst_lat32))
    assert worst < 1e-11 and worst_lat < 1e-11 and worst_lat32 < 2e-5
