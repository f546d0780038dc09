"""Physical-invariant checks of the oracle's physics step (the part no reference fixture
pins: "parity unpinned").  They hold for any correct implementation of the equations of
motion and the first-order integrator, independently of MuJoCo."""
import numpy as np
import pytest

RAW = np.array([1.0, 0.17, 7.0, 0.01, 1.2, 0.3])


def _model(orc, load, free=False):
    r = RAW.copy()
    if not load:
        r[4:] = 0
    m = orc.build_model(r)
    if free:
        m.density = m.viscosity = m.damping = m.gravity = 0.0
    return m


def _state(load, rng):
    nq, nv = (9, 8) if load else (7, 6)
    qpos = np.zeros(nq); qpos[2] = 15
    q = rng.normal(size=4); qpos[3:7] = q / np.linalg.norm(q)
    if load:
        qpos[7:] = [0.5, -0.4]
    return qpos, rng.normal(size=nv)


@pytest.mark.parametrize("load", [1, 0])
def test_mass_matrix_symmetric_positive_definite(orc, load):
    rng = np.random.default_rng(0)
    m = _model(orc, load)
    for _ in range(20):
        qpos, _ = _state(load, rng)
        M = orc.mass_matrix(m, qpos)
        np.testing.assert_allclose(M, M.T, atol=1e-14)
        assert np.linalg.eigvalsh(M).min() > 0
        mt = m.m0 + (m.m1 + m.m2 if load else 0)
        np.testing.assert_allclose(M[:3, :3], mt * np.eye(3), atol=1e-13)


@pytest.mark.parametrize("load", [1, 0])
def test_momentum_and_energy_conservation_first_order(orc, load):
    """no gravity / drag / damping / thrust: momenta and kinetic energy are conserved up to the O(h) error of
    the semi-implicit Euler scheme, i.e. the drift shrinks ~4x when h shrinks 4x"""
    rng = np.random.default_rng(1)
    m = _model(orc, load, free=True)
    qpos0, qvel0 = _state(load, rng)
    drift = []
    for h in (1e-3, 2.5e-4):
        qp, qv, act = qpos0.copy(), qvel0.copy(), np.zeros(4)
        ke0, _, l0, a0 = orc.energy_momentum(m, qp, qv)
        for _ in range(int(round(0.5 / h))):
            qp, qv, act, _ = orc.step(m, h, 1, qp, qv, act, np.zeros(4))
        ke, _, l, a = orc.energy_momentum(m, qp, qv)
        drift.append((abs(ke - ke0) / ke0, np.abs(l - l0).max(), np.abs(a - a0).max()))
    for d1, d2 in zip(*drift):
        assert d2 < d1 * 0.3 + 1e-12
    assert drift[1][0] < 1e-5 and drift[1][1] < 1e-4 and drift[1][2] < 1e-4


@pytest.mark.parametrize("load", [1, 0])
def test_free_fall_and_gravity_impulse(orc, load):
    rng = np.random.default_rng(2)
    m = _model(orc, load)
    m.density = m.viscosity = m.damping = 0.0
    nq, nv = (9, 8) if load else (7, 6)
    qp = np.zeros(nq); qp[2] = 15; qp[3] = 1
    qacc, _, sens = orc.forward(m, qp, np.zeros(nv), np.zeros(4), np.zeros(4))
    np.testing.assert_allclose(qacc[:3], [0, 0, -9.81], atol=1e-12)
    np.testing.assert_allclose(sens, 0, atol=1e-12)                       # an accelerometer in free fall reads zero
    qpos0, qvel0 = _state(load, rng)
    mt = m.m0 + (m.m1 + m.m2 if load else 0)
    h, n = 2.5e-4, 2000
    qp, qv, act = qpos0.copy(), qvel0.copy(), np.zeros(4)
    _, _, l0, _ = orc.energy_momentum(m, qp, qv)
    for _ in range(n):
        qp, qv, act, _ = orc.step(m, h, 1, qp, qv, act, np.zeros(4))
    _, _, l, _ = orc.energy_momentum(m, qp, qv)
    np.testing.assert_allclose(l - l0, [0, 0, -mt * 9.81 * h * n], atol=2e-4)


@pytest.mark.parametrize("load", [1, 0])
def test_hover_equilibrium_and_accelerometer(orc, load):
    m = _model(orc, load)
    nq, nv = (9, 8) if load else (7, 6)
    mt = m.m0 + (m.m1 + m.m2 if load else 0)
    a_h = mt * 9.81 / (4 * m.gearF)
    qp = np.zeros(nq); qp[2] = 15; qp[3] = 1
    qacc, act_dot, sens = orc.forward(m, qp, np.zeros(nv), np.full(4, a_h), np.full(4, a_h))
    assert np.abs(qacc).max() < 1e-12 and np.abs(act_dot).max() < 1e-12
    np.testing.assert_allclose(sens, [0, 0, 9.81], atol=1e-12)           # at rest the sensor reads +g


def test_actuator_filter_and_yaw_torque_signs(orc):
    m = _model(orc, 0)
    qp = np.zeros(7); qp[2] = 15; qp[3] = 1
    # activations follow ctrl with time constant tau, explicit Euler: act += h (ctrl - act) / tau
    _, _, act, _ = orc.step(m, 0.001, 1, qp, np.zeros(6), np.zeros(4), np.array([1, 0.5, 2.0, -1.0]))
    np.testing.assert_allclose(act, 0.001 / m.tau * np.array([1, 0.5, 1.0, 0.0]), atol=1e-15)  # ctrl clamped to [0,1]
    # rotor 0 only: positive yaw torque (gear +F/100), roll/pitch torque from its position (+x, -y)
    qacc, _, _ = orc.forward(m, qp, np.zeros(6), np.array([1.0, 0, 0, 0]), np.zeros(4))
    assert qacc[5] > 0 and qacc[3] < 0 and qacc[4] < 0
    # h / tau > 1 overshoots (reference quirk C-11): act leaves [0,1]
    _, _, act, _ = orc.step(m, 0.0133, 1, qp, np.zeros(6), np.ones(4), np.zeros(4))
    assert act.min() < 0


def test_model_matches_closed_forms(orc):
    """total mass, COM, and the tether stack of the default parameters, from env_gen.py by hand"""
    m = orc.build_model([1.35, 0.15, 7.5, 0.015, 1.2, 0.3])
    assert abs(m.m0 - (0.756 + 4 * 0.0945 + 4 * 0.054)) < 1e-12
    assert abs(m.c0[2] - 4 * 0.054 * 0.015 / m.m0) < 1e-15 and abs(m.c0[0]) < 1e-15 and abs(m.c0[1]) < 1e-15
    assert abs(m.m2 - (0.24 + 0.3)) < 1e-12
    assert abs(m.lc - (0.24 * 0.6 + 0.3 * 1.2) / 0.54) < 1e-12
    assert m.gearF == 7.5 and list(m.gearT) == [0.075, -0.075, 0.075, -0.075] and m.tau == 0.015
    # Ixx and Iyy differ by ~2.6e-9: the 5-digit Euler angles of arms 2 and 3 (2.3562, 3.927) are not exact
    assert 1e-10 < abs(m.I0full[0] - m.I0full[1]) < 1e-8 and m.I0full[2] > m.I0full[0]
    assert max(abs(m.I0full[3]), abs(m.I0full[4]), abs(m.I0full[5])) < 1e-12
    assert orc.round5g(0.123456789) == 0.12346 and orc.round5g(-1234.5678) == -1234.6 and orc.round5g(2e-5) == 2e-5


@pytest.mark.parametrize("axis", [0, 1])
def test_hanging_load_small_oscillation_matches_damped_pendulum(orc, axis):
    """Textbook anchor for the load dynamics: under a very heavy hovering drone the load is a damped physical pendulum,
    theta'' + (c / I) theta' + (m2 g lc / I) theta = 0 with I = I_com + m2 lc^2 about the hinge and c the hinge damping.
    The simulated period and logarithmic decrement must match the analytic ones (air drag switched off; the drone's recoil
    and counter-rotation are ~1e-5 with the masses used: with a 1000 kg drone the counter-rotation alone shifts the
    decrement by 6 %)."""
    r = RAW.copy(); r[0] = 1.0e6; r[2] = 5.0e6        # a 1000-tonne drone (it must not rotate either: I0 ~ 1e4 kg m^2)
    m = orc.build_model(r)
    m.density = m.viscosity = 0.0
    total = m.m0 + m.m1 + m.m2
    ctrl = np.full(4, total * abs(m.gravity) / (4 * m.gearF))
    assert np.all(ctrl < 1.0)
    qpos = np.zeros(9); qpos[2] = 15; qpos[3] = 1; qpos[7 + axis] = 0.05
    qvel = np.zeros(8); act = ctrl.copy()
    h, n = 0.002, 6000
    th = np.empty(n)
    for k in range(n):
        qpos, qvel, act, _ = orc.step(m, h, 1, qpos, qvel, act, ctrl)
        th[k] = qpos[7 + axis]
    # the outer hinge (about x) carries the link sphere and the pendulum, the inner one (about y) the pendulum only
    I = m.I2[axis] + m.m2 * m.lc ** 2 + (m.I1 if axis == 0 else 0.0)
    k_, c = m.m2 * abs(m.gravity) * m.lc, m.damping
    wd = np.sqrt(k_ / I - (c / (2 * I)) ** 2)
    up = np.where((th[:-1] < 0) & (th[1:] >= 0))[0]
    t_cross = (up + th[up] / (th[up] - th[up + 1])) * h          # interpolated upward zero crossings
    period = np.mean(np.diff(t_cross))
    assert abs(period - 2 * np.pi / wd) / (2 * np.pi / wd) < 2e-3
    peaks = [th[a:b].max() for a, b in zip(up[:-1], up[1:])]
    decrement = np.mean(np.log(np.array(peaks[:-1]) / np.array(peaks[1:])))
    assert abs(decrement - c / (2 * I) * 2 * np.pi / wd) / decrement < 2e-2
    assert abs(qpos[2] - 15) < 1e-3 and np.abs(qpos[:2]).max() < 5e-3     # the heavy drone stays put (mm of drift from the first swing)


# ------------------------------------------------------------------ floor contact: analytic anchors
def test_floor_rest_force_equals_weight_and_settles_critically_damped(orc):
    """a drone dropped from 3 cm with the rotors off lands on the four lower corners of its core box; at rest the normal force is
    the weight (to the solver tolerance), the penetration is sub-millimetre, and the approach to rest after the impact has
    MuJoCo's default constraint time constant: solref (0.02, 1) = critically damped, z - z_rest ~ (a + b t) exp(-t / 0.02 / dmax')"""
    m = _model(orc, 0)
    qpos = np.array([0.3, -0.2, 0.0166667 + 0.03, 1, 0, 0, 0.0]); qvel = np.zeros(6); act = np.zeros(4)
    zs, fs, ns = [], [], []
    for k in range(600):
        qpos, qvel, act, sens, n, fz = orc.step_floor(m, 0.001, 1, qpos, qvel, act, np.zeros(4))
        zs.append(qpos[2]); fs.append(fz); ns.append(n)
    w = m.m0 * m.gravity
    assert ns[-1] == 4 and abs(fs[-1] - w) < 1e-3 * w
    pen = 0.016667 - zs[-1]
    assert 0 < pen < 5e-4                                            # stiff: fractions of a millimetre
    assert abs(qvel).max() < 1e-4 and abs(qpos[0] - 0.3) < 1e-6 and abs(qpos[1] + 0.2) < 1e-6
    # accelerometer at rest on the floor reads +g along body z (specific force)
    np.testing.assert_allclose(sens, [0, 0, m.gravity], atol=2e-3)
    # after the first rebound the height error decays monotonically (no oscillation: damping ratio 1)
    tail = np.array(zs[200:]) - zs[-1]
    assert np.all(np.diff(np.abs(tail)) <= 1e-9)


def test_floor_friction_is_coulomb_with_mu_one(orc):
    """sliding on the floor: a level drone given a horizontal velocity decelerates at nearly mu g = g until it sticks (pyramidal cone,
    friction 1 from max(geom friction 1, floor friction 1)); a slow push below the cone limit does not move it"""
    m = _model(orc, 0)
    qpos = np.array([0, 0, 0.01655, 1, 0, 0, 0.0]); qvel = np.zeros(6); act = np.zeros(4)
    for _ in range(300):
        qpos, qvel, act, _, _, _ = orc.step_floor(m, 0.001, 1, qpos, qvel, act, np.zeros(4))
    qvel[0] = 1.0                                                    # 1 m/s along x
    v = []
    for _ in range(60):
        qpos, qvel, act, _, n, fz = orc.step_floor(m, 0.001, 1, qpos, qvel, act, np.zeros(4))
        v.append(qvel[0])
    dec = -(v[50] - v[10]) / 0.040
    # an elliptic cone would give exactly mu g; in the pyramidal cone the two edges across the sliding direction keep carrying part
    # of the normal load without contributing friction, so the deceleration sits a little below g (9.07 m/s^2 with these constants)
    assert 0.88 * m.gravity < dec < 1.0 * m.gravity, dec
    for _ in range(400):
        qpos, qvel, act, _, _, _ = orc.step_floor(m, 0.001, 1, qpos, qvel, act, np.zeros(4))
    assert abs(qvel[:3]).max() < 1e-3                                # stuck


def test_floor_is_inactive_in_flight(orc):
    """above the floor the step with contact IS the step without (no geom below z = 0): bit-identical states"""
    rng = np.random.default_rng(8)
    for load in (0, 1):
        m = _model(orc, load)
        nq, nv = (9, 8) if load else (7, 6)
        qpos = np.zeros(nq); qpos[:3] = [0, 0, 3.0]; q = rng.normal(size=4); qpos[3:7] = q / np.linalg.norm(q)
        qvel = rng.normal(size=nv); act = rng.uniform(0, 1, 4); ctrl = rng.uniform(0, 1, 4)
        a = orc.step(m, 0.01, 3, qpos, qvel, act, ctrl)
        b = orc.step_floor(m, 0.01, 3, qpos, qvel, act, ctrl)
        for x, y in zip(a, b[:4]):
            np.testing.assert_array_equal(x, y)
        assert b[4] == 0


def test_floor_contact_of_the_hanging_load(orc):
    """load model: the 1.2 m tether lets the load box reach the floor while the drone hovers at 1 m: the box rests on the floor
    (contacts on body 2), the tether goes slack-free (rigid rod) and the normal force carries part of the weight"""
    m = _model(orc, 1)
    mt = m.m0 + m.m1 + m.m2
    qpos = np.array([0, 0, 1.0, 1, 0, 0, 0, 0.3, 0.0]); qvel = np.zeros(8)
    hover = mt * m.gravity / (4 * m.gearF)
    act = np.full(4, hover)
    seen = 0
    for k in range(400):
        qpos, qvel, act, sens, n, fz = orc.step_floor(m, 0.002, 1, qpos, qvel, act, np.full(4, hover))
        seen = max(seen, n)
        assert np.all(np.isfinite(qpos)) and fz >= 0
    cons = orc.floor_contacts(m, qpos)
    assert seen > 0 and all(b == 2 for _, _, b in cons)


def test_torque_free_precession_of_the_airframe(orc):
    """Textbook anchor for the gyroscopic term: a free symmetric top (Ix = Iy) spinning about z with a small transverse rate
    precesses in the body frame, wx + i wy = eps exp(i Omega t) with Omega = (Iz - Ix) / Ix * wz.  No gravity, air or thrust."""
    m = _model(orc, 0)
    m.density = m.viscosity = 0.0
    m.gravity = 0.0
    Ix, Iy, Iz = m.I0full[0], m.I0full[1], m.I0full[2]
    assert abs(Ix - Iy) < 1e-3 * Ix                      # the four arms at +-45 degrees make it a symmetric top (to the XML's rounding)
    wz, eps, h, n = 25.0, 0.05, 2e-5, 20000                # explicit Euler grows the amplitude by (Omega h)^2 / 2 per step: 0.2 % here
    qpos = np.array([0, 0, 5, 1, 0, 0, 0.0]); qvel = np.array([0, 0, 0, eps, 0, wz]); act = np.zeros(4)
    Omega = (Iz - 0.5 * (Ix + Iy)) / (0.5 * (Ix + Iy)) * wz
    for k in range(n):
        qpos, qvel, act, _ = orc.step(m, h, 1, qpos, qvel, act, np.zeros(4))
    t = n * h
    want = eps * np.exp(1j * Omega * t)
    got = qvel[3] + 1j * qvel[4]
    assert abs(got - want) < 5e-3 * eps, (got, want, Omega)
    assert abs(np.angle(got / want)) < 1e-3                # the precession phase itself: Omega t to a milliradian
    assert abs(qvel[5] - wz) < 1e-9 * wz
