"""ctypes front-end of the CPU oracle (oracle/qd_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

OBS_KINDS = [
    "raw", "GlobalFrameRPYEnv", "LocalFramePRYEnv", "LocalFrameFullStateEnv", "LocalFrameFullStateZvecEnv",
    "LocalFramePRYaccEnv", "LocalFramePRYParamsEnv", "LocalFramePRYaccParamsEnv", "LocalFrameRPYParamsEnv",
    "LocalFrameRPYFakeParamsEnv", "LocalFrameRPYEnv", "LocalFramePRYaccNoPendEnv",
    "LocalFramePRYaccParamsNoPendEnv", "LocalFrameRmParamsEnv", "LocalFrameZvecEnv", "SimpleDrone",
]
REWARD_KINDS = [
    "default_reward_fcn", "distance_reward_fcn", "distance_energy_reward",
    "distance_energy_reward_pendulum_angle", "distance_energy_reward_pendulum_angle2",
    "distance_energy_reward_pendulum_angle3", "distance_energy_reward_pendulum_en",
    "distance_energy_reward_pendulum_en2", "distance_energy_reward_pendulum_en3",
    "distance_energy_reward_pendulum_en4", "distance_time_energy_reward", "reward_1", "reward_pendulum_dist",
    "reward_pendulumDistHeading", "reward_2", "reward_2_penergy", "reward_3", "simple_drone",
]


class OrcModel(C.Structure):
    _fields_ = [
        ("load", C.c_int), ("gravity", C.c_double), ("density", C.c_double), ("viscosity", C.c_double),
        ("damping", C.c_double), ("m0", C.c_double), ("c0", C.c_double * 3), ("I0full", C.c_double * 6),
        ("I0", C.c_double * 3), ("R0i", C.c_double * 9), ("box0", C.c_double * 3),
        ("rotor", (C.c_double * 3) * 4), ("gearF", C.c_double), ("gearT", C.c_double * 4), ("tau", C.c_double),
        ("sense", C.c_double * 3), ("anchor", C.c_double * 3), ("m1", C.c_double), ("I1", C.c_double),
        ("box1", C.c_double), ("m2", C.c_double), ("lc", C.c_double), ("I2", C.c_double * 3),
        ("box2", C.c_double * 3), ("raw", C.c_double * 6), ("invweight", (C.c_double * 2) * 3),
    ]


class OrcSampleCfg(C.Structure):
    _fields_ = [
        ("load", C.c_int), ("random_start", C.c_int), ("start_pos", C.c_double * 4),
        ("max_pos_offset", C.c_double), ("angle_var", C.c_double * 2), ("vel_var", C.c_double * 3),
        ("ang_vel_var", C.c_double * 3), ("pend_rp_var", C.c_double * 2), ("pend_vel_var", C.c_double * 2),
    ]


class OrcBatchCfg(C.Structure):
    _fields_ = [
        ("n", C.c_int), ("load", C.c_int), ("obs_kind", C.c_int), ("reward_kind", C.c_int),
        ("frame_skip", C.c_int), ("ctrl_map", C.c_int), ("max_steps", C.c_long), ("h", C.c_double),
        ("max_distance", C.c_double), ("ref", C.c_double * 4),
    ]


_lib = None
_lib_omp = None
_dp = C.POINTER(C.c_double)


def build(force=False):
    """gcc build of the oracle (plain and OpenMP variants) into oracle/_build/."""
    so = os.path.join(_BUILD, "libqd_oracle.so")
    src = os.path.join(_HERE, "qd_oracle.c")
    hdr = os.path.join(_HERE, "qd_oracle.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return so


def _p(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _setup(lib):
    lib.orc_round5g.restype = C.c_double
    lib.orc_round5g.argtypes = [C.c_double]
    lib.orc_reward.restype = C.c_double
    lib.orc_reward.argtypes = [C.c_int, _dp, C.c_int, _dp, C.c_long, _dp, C.c_double]
    lib.orc_truncated.restype = C.c_int
    lib.orc_truncated.argtypes = [_dp, _dp, C.c_long, C.c_double, C.c_long]
    lib.orc_obs.restype = C.c_int
    lib.orc_obs.argtypes = [C.c_int, _dp, C.c_int, _dp, _dp]
    lib.orc_obs_dim.restype = C.c_int
    lib.orc_obs_dim.argtypes = [C.c_int, C.c_int]
    lib.orc_drone_state.restype = C.c_int
    lib.orc_step.argtypes = [C.POINTER(OrcModel), C.c_double, C.c_int, _dp, _dp, _dp, _dp, _dp]
    lib.orc_step_floor.argtypes = [C.POINTER(OrcModel), C.c_double, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]
    lib.orc_step_floor.restype = C.c_int
    lib.orc_forward_floor.argtypes = [C.POINTER(OrcModel), _dp, _dp, _dp, C.c_double, _dp, _dp]
    lib.orc_forward_floor.restype = C.c_int
    lib.orc_floor_contacts.argtypes = [C.POINTER(OrcModel), _dp, C.c_void_p]
    lib.orc_floor_contacts.restype = C.c_int
    lib.orc_step.restype = None
    lib.orc_forward.argtypes = [C.POINTER(OrcModel), _dp, _dp, _dp, _dp, _dp, _dp, _dp]
    lib.orc_forward.restype = None
    lib.orc_gen_params_philox.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, _dp, _dp, C.c_double, C.c_int,
                                          C.c_int, _dp]
    lib.orc_gen_params_philox.restype = None
    lib.orc_sample_draws_philox.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_float),
                                            C.POINTER(C.c_float)]
    lib.orc_sample_draws_philox.restype = None
    for f in (lib.orc_pid_reset, lib.orc_pid_position, lib.orc_pid_tilts2rpy, lib.orc_pid_attitude,
              lib.orc_pid_action):
        f.restype = None
    return lib


def lib(omp=False):
    global _lib, _lib_omp
    build()
    if omp:
        if _lib_omp is None:
            _lib_omp = _setup(C.CDLL(os.path.join(_BUILD, "libqd_oracle_omp.so")))
        return _lib_omp
    if _lib is None:
        _lib = _setup(C.CDLL(os.path.join(_BUILD, "libqd_oracle.so")))
    return _lib


# ------------------------------------------------------------------ wrappers
def round5g(x):
    return lib().orc_round5g(float(x))


def build_model(raw):
    m = OrcModel()
    raw = _f64(raw)
    lib().orc_build_model(_p(raw), C.byref(m))
    return m


def build_models(raw):
    raw = _f64(raw).reshape(-1, 6)
    arr = (OrcModel * len(raw))()
    for i in range(len(raw)):
        lib().orc_build_model(_p(raw[i]), C.byref(arr[i]))
    return arr


def forward(model, qpos, qvel, act, ctrl):
    qpos, qvel, act, ctrl = _f64(qpos), _f64(qvel), _f64(act), _f64(ctrl)
    qacc = np.zeros(len(qvel))
    act_dot = np.zeros(4)
    sensor = np.zeros(3)
    lib().orc_forward(C.byref(model), _p(qpos), _p(qvel), _p(act), _p(ctrl), _p(qacc), _p(act_dot), _p(sensor))
    return qacc, act_dot, sensor


def step(model, h, nstep, qpos, qvel, act, ctrl):
    """returns new (qpos, qvel, act, sensor); inputs are not modified"""
    qpos, qvel, act, ctrl = _f64(qpos).copy(), _f64(qvel).copy(), _f64(act).copy(), _f64(ctrl)
    sensor = np.zeros(3)
    lib().orc_step(C.byref(model), float(h), int(nstep), _p(qpos), _p(qvel), _p(act), _p(ctrl), _p(sensor))
    return qpos, qvel, act, sensor


class OrcContact(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("dist", C.c_double), ("body", C.c_int)]


def floor_contacts(model, qpos):
    """contacts of the drone's geoms with the floor plane z = 0: list of (world position, signed distance, body)"""
    qpos = _f64(qpos)
    buf = (OrcContact * 64)()
    n = lib().orc_floor_contacts(C.byref(model), _p(qpos), C.cast(buf, C.c_void_p))
    return [(np.array(buf[i].pos[:]), buf[i].dist, buf[i].body) for i in range(n)]


def forward_floor(model, qpos, qvel, act, h):
    """qacc including the floor's reaction, number of contacts, normal force"""
    qpos, qvel, act = _f64(qpos), _f64(qvel), _f64(act)
    qacc, fz = np.zeros(8 if model.load else 6), np.zeros(1)
    n = lib().orc_forward_floor(C.byref(model), _p(qpos), _p(qvel), _p(act), float(h), _p(qacc), _p(fz))
    return qacc, n, float(fz[0])


def step_floor(model, h, nstep, qpos, qvel, act, ctrl):
    """orc_step with the floor contact: returns new (qpos, qvel, act, sensor, n_contacts, normal force)"""
    qpos, qvel, act, ctrl = _f64(qpos).copy(), _f64(qvel).copy(), _f64(act).copy(), _f64(ctrl)
    sensor, fz = np.zeros(3), np.zeros(1)
    n = lib().orc_step_floor(C.byref(model), float(h), int(nstep), _p(qpos), _p(qvel), _p(act), _p(ctrl), _p(sensor), _p(fz))
    return qpos, qvel, act, sensor, n, float(fz[0])


def mass_matrix(model, qpos):
    nv = 8 if model.load else 6
    M = np.zeros((nv, nv))
    qpos = _f64(qpos)
    lib().orc_mass_matrix(C.byref(model), _p(qpos), _p(M))
    return M


def energy_momentum(model, qpos, qvel):
    ke, pe = C.c_double(), C.c_double()
    lin, ang = np.zeros(3), np.zeros(3)
    qpos, qvel = _f64(qpos), _f64(qvel)
    lib().orc_energy_momentum(C.byref(model), _p(qpos), _p(qvel), C.byref(ke), C.byref(pe), _p(lin), _p(ang))
    return ke.value, pe.value, lin, ang


def quat2rpy(q):
    q = _f64(q); o = np.zeros(3); lib().orc_quat2rpy(_p(q), _p(o)); return o


def rpy2quat(r):
    r = _f64(r); o = np.zeros(4); lib().orc_rpy2quat(_p(r), _p(o)); return o


def quat2dcm(q):
    q = _f64(q); o = np.zeros(9); lib().orc_quat2dcm(_p(q), _p(o)); return o.reshape(3, 3)


def dcm2quat(R):
    R = _f64(R).reshape(9); o = np.zeros(4); lib().orc_dcm2quat(_p(R), _p(o)); return o


def pendrp2quat(rp):
    rp = _f64(rp); o = np.zeros(4); lib().orc_pendrp2quat(_p(rp), _p(o)); return o


def drone_state(load, qpos, qvel, sensor, act, ref, raw):
    o = np.zeros(33)
    a = [_f64(x) for x in (qpos, qvel, sensor, act, ref, raw)]
    n = lib().orc_drone_state(int(load), *[_p(x) for x in a], _p(o))
    return o[:n]


def obs(kind, s, ref):
    s, ref = _f64(s), _f64(ref)
    o = np.zeros(40)
    n = lib().orc_obs(int(kind), _p(s), len(s), _p(ref), _p(o))
    if n < 0:
        raise NameError("observation variant raises in the reference")
    return o[:n]


def obs_dim(kind, ns):
    return lib().orc_obs_dim(int(kind), int(ns))


def simple_obs(qpos):
    qpos = _f64(qpos); o = np.zeros(6); lib().orc_simple_obs(_p(qpos), _p(o)); return o


def reward(kind, s, action, num_steps, ref, max_distance):
    s, action, ref = _f64(s), _f64(action), _f64(ref)
    return lib().orc_reward(int(kind), _p(s), len(s), _p(action), int(num_steps), _p(ref), float(max_distance))


def truncated(s, ref, num_steps, max_distance, max_steps):
    s, ref = _f64(s), _f64(ref)
    return bool(lib().orc_truncated(_p(s), _p(ref), int(num_steps), float(max_distance), int(max_steps)))


def philox4x32(ctr, key):
    c = (C.c_uint32 * 4)(*[int(x) & 0xFFFFFFFF for x in ctr])
    k = (C.c_uint32 * 2)(*[int(x) & 0xFFFFFFFF for x in key])
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32(c, k, o)
    return [int(x) for x in o]


def sample_cfg(load, random_start, start_pos, max_pos_offset, angle_var, vel_var, ang_vel_var, pend_rp_var,
               pend_vel_var):
    c = OrcSampleCfg()
    c.load, c.random_start = int(load), int(random_start)
    c.start_pos[:] = [float(x) for x in start_pos]
    c.max_pos_offset = float(max_pos_offset)
    c.angle_var[:] = [float(x) for x in angle_var]
    c.vel_var[:] = [float(x) for x in vel_var]
    c.ang_vel_var[:] = [float(x) for x in ang_vel_var]
    c.pend_rp_var[:] = [float(x) for x in pend_rp_var]
    c.pend_vel_var[:] = [float(x) for x in pend_vel_var]
    return c


def sample_state_from_draws(cfg, z, u):
    z, u = _f64(z), _f64(u)
    qpos, qvel = np.zeros(9 if cfg.load else 7), np.zeros(8 if cfg.load else 6)
    lib().orc_sample_state_from_draws(C.byref(cfg), _p(z), _p(u), _p(qpos), _p(qvel))
    return qpos, qvel


def sample_draws_philox(seed, env, episode):
    z = np.zeros(15, dtype=np.float32)
    u = np.zeros(2, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    lib().orc_sample_draws_philox(int(seed), int(env), int(episode), z.ctypes.data_as(fp), u.ctypes.data_as(fp))
    return z, u


def sample_state_philox(cfg, seed, env, episode):
    z, u = sample_draws_philox(seed, env, episode)
    return sample_state_from_draws(cfg, z.astype(np.float64), u.astype(np.float64))


def gen_params_philox(seed, env, regen, center, width, difficulty, random_params, load):
    center, width = _f64(center), _f64(width)
    raw = np.zeros(6)
    lib().orc_gen_params_philox(int(seed), int(env), int(regen), _p(center), _p(width), float(difficulty),
                                int(bool(random_params)), int(bool(load)), _p(raw))
    return raw


def trajectory(mode, p, start, end, dt, n):
    """first n samples of gen_{circle,step,ramp}_trajectory (evaluation.py:135-152); mode 1 / 2 / 3"""
    pp = np.zeros(4); pp[:len(p)] = p
    start, end, out = _f64(start), _f64(end), np.zeros((n, 4))
    f = lib().orc_trajectory_point
    f.restype = None
    for k in range(n):
        f(int(mode), _p(pp), _p(start), _p(end), C.c_double(dt), C.c_long(k), _p(out[k]))
    return out


class OrcPid(C.Structure):
    _fields_ = [("pos_i", C.c_double * 3), ("pos_prev", C.c_double * 3), ("att_i", C.c_double * 3),
                ("att_prev", C.c_double * 3), ("pos_first", C.c_int), ("att_first", C.c_int)]


class Pid:
    """The reference's PositionController + AttittudeController pair for n drones
    (models/Analytic/*.py, driven as attitude_test.py:36-47)."""

    def __init__(self, masses, forces):
        self.masses, self.forces = _f64(masses).ravel(), _f64(forces).ravel()
        self.n = len(self.masses)
        self.c = (OrcPid * self.n)()
        for i in range(self.n):
            lib().orc_pid_reset(C.byref(self.c[i]))

    def position(self, ref, xyz):
        ref, xyz, out = _f64(ref), _f64(xyz).reshape(self.n, 3), np.zeros((self.n, 3))
        for i in range(self.n):
            lib().orc_pid_position(C.byref(self.c[i]), _p(ref), _p(xyz[i]), _p(out[i]))
        return out

    @staticmethod
    def tilts2rpy(pos_action, heading):
        pa = _f64(pos_action).reshape(-1, 3)
        out = np.zeros((len(pa), 4))
        for i in range(len(pa)):
            lib().orc_pid_tilts2rpy(_p(pa[i]), C.c_double(float(heading)), _p(out[i]))
        return out

    def attitude(self, rpyz, rpy):
        rpyz, rpy, out = _f64(rpyz).reshape(self.n, 4), _f64(rpy).reshape(self.n, 3), np.zeros((self.n, 4))
        for i in range(self.n):
            lib().orc_pid_attitude(C.byref(self.c[i]), _p(rpyz[i]), _p(rpy[i]), C.c_double(self.masses[i]),
                                   C.c_double(self.forces[i]), _p(out[i]))
        return out

    def action(self, ref, xyz, rpy):
        """state -> env action, clip(ctrl - 0.1, 0, 1) included (attitude_test.py:47)"""
        ref, xyz, rpy = _f64(ref), _f64(xyz).reshape(self.n, 3), _f64(rpy).reshape(self.n, 3)
        out = np.zeros((self.n, 4))
        for i in range(self.n):
            lib().orc_pid_action(C.byref(self.c[i]), _p(ref), _p(xyz[i]), _p(rpy[i]), C.c_double(self.masses[i]),
                                 C.c_double(self.forces[i]), _p(out[i]))
        return out


class Batch:
    """N independent drones stepped on the CPU (the bench's cpu_baseline and the
    multi-step parity tests).  AoS float64 state."""

    def __init__(self, raw, load, obs_kind, reward_kind, h, frame_skip, ctrl_map, ref, max_distance, max_steps):
        self.raw = _f64(raw).reshape(-1, 6).copy()
        self.n = len(self.raw)
        self.models = build_models(self.raw)
        self.load = int(load)
        nq, nv = (9, 8) if load else (7, 6)
        self.qpos = np.zeros((self.n, nq)); self.qpos[:, 3] = 1.0
        self.qvel = np.zeros((self.n, nv))
        self.act = np.zeros((self.n, 4))
        self.sensor = np.zeros((self.n, 3))
        self.num_steps = np.zeros(self.n, dtype=np.int64)
        c = OrcBatchCfg()
        c.n, c.load, c.obs_kind, c.reward_kind = self.n, self.load, int(obs_kind), int(reward_kind)
        c.frame_skip, c.ctrl_map, c.max_steps = int(frame_skip), int(ctrl_map), int(max_steps)
        c.h, c.max_distance = float(h), float(max_distance)
        c.ref[:] = [float(x) for x in ref]
        self.cfg = c
        self.D = obs_dim(obs_kind, 33 if load else 29)
        self.obs = np.zeros((self.n, self.D))
        self.reward = np.zeros(self.n)
        self.trunc = np.zeros(self.n, dtype=np.uint8)

    def step(self, actions, threads=1):
        actions = _f64(actions).reshape(self.n, 4)
        L = lib(omp=threads > 1)
        L.orc_batch_step(C.byref(self.cfg), self.models, _p(self.raw), _p(self.qpos), _p(self.qvel), _p(self.act),
                         _p(self.sensor), self.num_steps.ctypes.data_as(C.POINTER(C.c_long)), _p(actions),
                         _p(self.obs), _p(self.reward), self.trunc.ctypes.data_as(C.POINTER(C.c_ubyte)),
                         int(threads))
        return self.obs, self.reward, self.trunc
