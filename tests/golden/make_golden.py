#!/usr/bin/env python3
"""Generate tests/golden/reference_vectors.npz from the reference's own Python.

Runs ONLY in the build container (needs /root/reference); never on the GPU box.
The reference's numpy/scipy-only code (transformation.py, rewards.py,
observation_wrappers.py, BaseDroneEnv.{sample_state, generate_drone_params,
get_drone_states, default_termination_fcn}, SimpleDrone.{_get_obs, step}) is
imported from where it lies and executed on seeded inputs; the inputs and the
outputs are stored as data.  Third-party modules the reference imports at module
scope but that are absent here (gymnasium, ray, mujoco, dm_control, glfw,
pygame) are replaced by empty placeholder modules in sys.modules just so the
import statements succeed -- none of their functionality is used by the
functions executed here (the physics, mujoco.mj_step, cannot be run and is NOT
covered by these vectors).

Nothing from /root/reference is copied: the output holds numbers only.
"""
import json
import os
import sys
import types
import typing

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.npz")


def _install_placeholders():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    T = typing.TypeVar("T")

    class Space(typing.Generic[T]):
        pass

    class Box(Space):
        def __init__(self, low=None, high=None, shape=None, dtype=None, seed=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class EzPickle:
        def __init__(self, *a, **k):
            pass

    class VectorEnv:
        def __init__(self, *a, **k):
            pass

    class Viewer:
        pass

    seeding = mod("gymnasium.utils.seeding", np_random=lambda seed=None: (np.random.default_rng(seed), seed))
    utils = mod("gymnasium.utils", EzPickle=EzPickle, seeding=seeding)
    spaces = mod("gymnasium.spaces", Space=Space, Box=Box)
    error = mod("gymnasium.error", DependencyNotInstalled=type("DependencyNotInstalled", (Exception,), {}),
                Error=Exception)
    mod("gymnasium", utils=utils, spaces=spaces, error=error, logger=types.SimpleNamespace(warn=print), Env=object)
    mod("gymnasium.envs")
    mod("gymnasium.envs.mujoco")
    mod("gymnasium.envs.mujoco.mujoco_rendering", Viewer=Viewer)
    mod("ray"); mod("ray.rllib"); mod("ray.rllib.env")
    mod("ray.rllib.env.vector_env", VectorEnv=VectorEnv)
    mod("mujoco"); mod("glfw"); mod("pygame")
    mod("dm_control", mjcf=mod("dm_control.mjcf"), mujoco=mod("dm_control.mujoco"))


class RecordingRNG:
    """np.random.Generator front that records the raw standard-normal / uniform
    draws the reference consumes, in order."""

    def __init__(self, seed):
        self.g = np.random.default_rng(seed)
        self.z, self.u = [], []

    def normal(self, loc=0.0, scale=1.0, size=None):
        if size is None:
            size = np.shape(scale)
        z = self.g.standard_normal(size)
        self.z.extend(np.ravel(z).tolist())
        return loc + np.asarray(scale) * z

    def random(self, size=None):
        u = self.g.random(size)
        self.u.extend(np.ravel(u).tolist())
        return u

    def uniform(self, low=0.0, high=1.0, size=None):
        u = self.g.random(size)
        self.u.extend(np.ravel(u).tolist())
        return low + (high - low) * u


def main():
    _install_placeholders()
    sys.path.insert(0, REF)
    import environments.transformation as tr
    import environments.rewards as rw
    import environments.BaseDroneEnv as bde
    import environments.observation_wrappers as ow
    import environments.SimpleDrone as sd

    rng = np.random.default_rng(20250614)
    out = {}

    # ---------------------------------------------------------------- a5
    n = 64
    quats = rng.normal(size=(n, 4))
    quats /= np.linalg.norm(quats, axis=1, keepdims=True)
    quats[0] = [1, 0, 0, 0]
    quats[1] = [0.70710678, 0, 0.70710678, 0]  # pitch = +90 deg (gimbal lock)
    rpys = np.column_stack([rng.uniform(-np.pi, np.pi, n), rng.uniform(-1.5, 1.5, n), rng.uniform(-np.pi, np.pi, n)])
    rpys[0] = [0.1, -0.2, 0.3]
    prps = rng.uniform(-1.2, 1.2, size=(n, 2))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out["tr_quat"] = quats
        out["tr_quat2rpy"] = np.array([tr.mujoco_quat2rpy(q) for q in quats])
        out["tr_quat2dcm"] = np.array([tr.mujoco_quat2DCM(q) for q in quats])
        out["tr_rpy"] = rpys
        out["tr_rpy2quat"] = np.array([tr.mujoco_rpy2quat(r) for r in rpys])
        out["tr_dcm2quat"] = np.array([tr.mujoco_DCM2quat(tr.mujoco_quat2DCM(q)) for q in quats])
        out["tr_prp"] = prps
        out["tr_pendrp2quat"] = np.array([tr.mujoco_pendulumrp2quat(p) for p in prps])

    # ------------------------------------------------- states for obs / rewards
    def rand_states(n, ns):
        s = np.zeros((n, ns))
        s[:, 0:3] = np.array([0, 0, 15]) + rng.normal(scale=1.5, size=(n, 3))
        s[:, 3:6] = np.column_stack([rng.uniform(-1.2, 1.2, n), rng.uniform(-1.2, 1.2, n),
                                     rng.uniform(-np.pi, np.pi, n)])
        s[:, 6:12] = rng.normal(scale=1.5, size=(n, 6))
        s[:, 12:ns - 10] = rng.normal(scale=0.7, size=(n, ns - 22))  # pend rp/vel (load), acc, act
        s[:, ns - 14:ns - 10] = rng.uniform(0, 1, size=(n, 4))       # act
        s[:, ns - 6:] = np.array([1, 0.17, 7, 0.01, 1.2, 0.3]) * rng.uniform(0.8, 1.2, size=(n, 6))
        return s

    n = 48
    ref = np.array([0.3, -0.2, 15.0, 0.7])
    S33 = rand_states(n, 33)
    S33[:, 23:27] = ref
    S33[0, 0:3] = ref[:3] + [0.05, 0.02, -0.03]   # close to the reference (reward branches)
    S33[1, 0:3] = ref[:3] + [3.0, 3.0, 1.0]       # beyond max_distance
    S29 = rand_states(n, 29)
    S29[:, 19:23] = ref
    S29[:, 27:29] = 0.0                            # pendulum*value = 0 without load
    actions = rng.uniform(0, 1, size=(n, 4))
    num_steps = rng.integers(0, 1100, size=n)
    num_steps[2] = 512
    max_distance, max_steps = 4.0, 512
    out["st33"], out["st29"], out["st_ref"] = S33, S29, ref
    out["st_actions"], out["st_num_steps"] = actions, num_steps
    out["st_max_distance"], out["st_max_steps"] = np.float64(max_distance), np.int64(max_steps)

    env = types.SimpleNamespace(reference=ref, max_distance=max_distance, max_steps=max_steps)
    reward_names = ["default_reward_fcn", "distance_reward_fcn", "distance_energy_reward",
                    "distance_energy_reward_pendulum_angle", "distance_energy_reward_pendulum_angle2",
                    "distance_energy_reward_pendulum_angle3", "distance_energy_reward_pendulum_en",
                    "distance_energy_reward_pendulum_en2", "distance_energy_reward_pendulum_en3",
                    "distance_energy_reward_pendulum_en4", "distance_time_energy_reward", "reward_1",
                    "reward_pendulum_dist", "reward_pendulumDistHeading", "reward_2", "reward_2_penergy",
                    "reward_3"]
    for name in reward_names:
        f = getattr(rw, name)
        out["rew_" + name] = np.array([float(f(env, S33[i], actions[i], num_steps[i])) for i in range(n)])
    out["trunc33"] = np.array([bool(bde.default_termination_fcn(env, S33[i], actions[i], num_steps[i]))
                               for i in range(n)])
    # rewards that do not index params also run on the 29-vector (no-load quirk C-7)
    for name in reward_names[:6] + ["distance_time_energy_reward", "reward_1"]:
        f = getattr(rw, name)
        out["rew29_" + name] = np.array([float(f(env, S29[i], actions[i], num_steps[i])) for i in range(n)])

    # ------------------------------------------------------------------ a8
    obs_classes = ["GlobalFrameRPYEnv", "LocalFramePRYEnv", "LocalFrameFullStateEnv", "LocalFrameFullStateZvecEnv",
                   "LocalFramePRYaccEnv", "LocalFramePRYParamsEnv", "LocalFramePRYaccParamsEnv",
                   "LocalFrameRPYParamsEnv", "LocalFrameRPYFakeParamsEnv", "LocalFrameRPYEnv",
                   "LocalFramePRYaccNoPendEnv", "LocalFrameRmParamsEnv", "LocalFrameZvecEnv"]
    for name in obs_classes:
        cls = getattr(ow, name)
        for tag, S in (("33", S33), ("29", S29)):
            e = object.__new__(cls)
            e.states = [S[i] for i in range(n)]
            e.reference = ref
            out["obs%s_%s" % (tag, name)] = np.array(e._get_obs())
    try:
        e = object.__new__(ow.LocalFramePRYaccParamsNoPendEnv)
        e.states = [S33[0]]
        e.reference = ref
        e._get_obs()
        broken = "no error"
    except NameError as ex:
        broken = "NameError"
    out["obs_broken_variant_error"] = np.array(broken)

    # ------------------------------------------------------------------ a4
    for load in (1, 0):
        nd = 5
        nq, nv = (9, 8) if load else (7, 6)
        qpos = rng.normal(size=nd * nq)
        for i in range(nd):
            q = qpos[nq * i + 3:nq * i + 7]
            qpos[nq * i + 3:nq * i + 7] = q / np.linalg.norm(q)
        qvel = rng.normal(size=nd * nv)
        sens = rng.normal(size=nd * 3)
        act = rng.uniform(0, 1, size=nd * 4)
        params = [dict(mass=1.0 + 0.01 * i, arm_len=0.17, motor_force=7.0, motor_tau=0.01,
                       pendulum_len=load * 1.2, weight_mass=load * 0.3) for i in range(nd)]
        e = object.__new__(bde.BaseDroneEnv)
        e.pendulum, e.num_drones, e.reference, e.drone_params = bool(load), nd, ref, params
        e.data = types.SimpleNamespace(qpos=qpos, qvel=qvel, sensordata=sens, act=act)
        st = np.array(e.get_drone_states())
        k = "gds%d_" % load
        out[k + "qpos"], out[k + "qvel"], out[k + "sens"], out[k + "act"] = qpos, qvel, sens, act
        out[k + "params"] = np.array([list(p.values()) for p in params])
        out[k + "states"] = st

    # ------------------------------------------------------------------ a9
    cases = []
    base = bde.base_config
    case_cfgs = [
        dict(pendulum=True, random_start_pos=True, state_difficulty=0.4, max_random_offset=2, angle_variance=[0, 0],
             vel_variance=[1, 1, 1], ang_vel_variance=[1, 1, 1], pendulum_rp_variance=[0.5, 0.5],
             pendulum_ang_vel_variance=[0.5, 0.5], start_pos=[0, 0, 15, 0]),
        dict(pendulum=True, random_start_pos=True, state_difficulty=1.0, max_random_offset=2,
             angle_variance=[0.8, 0.8], vel_variance=[1, 2, 3], ang_vel_variance=[0.5, 1, 1.5],
             pendulum_rp_variance=[0.5, 0.25], pendulum_ang_vel_variance=[0.5, 1.0], start_pos=[1, -2, 10, 0.5]),
        dict(pendulum=False, random_start_pos=True, state_difficulty=0.8, max_random_offset=1,
             angle_variance=[0.3, 0.3], vel_variance=[1, 1, 1], ang_vel_variance=[1, 1, 1],
             pendulum_rp_variance=[0.5, 0.5], pendulum_ang_vel_variance=[0.5, 0.5], start_pos=[0, 0, 15, 0]),
        dict(pendulum=True, random_start_pos=False, state_difficulty=0.4, max_random_offset=2, angle_variance=[0, 0],
             vel_variance=[1, 1, 1], ang_vel_variance=[1, 1, 1], pendulum_rp_variance=[0.5, 0.5],
             pendulum_ang_vel_variance=[0.5, 0.5], start_pos=[0.5, 0.25, 12, -1.0]),
    ]
    for ci, cc in enumerate(case_cfgs):
        e = object.__new__(bde.BaseDroneEnv)
        sdiff = cc["state_difficulty"]
        e.pendulum = cc["pendulum"]
        e.random_start_pos = cc["random_start_pos"]
        e.start_pos = cc["start_pos"]
        e.max_pos_offset = sdiff * cc["max_random_offset"]
        e.angle_variance = sdiff * np.array(cc["angle_variance"])
        e.ang_vel_variance = sdiff * np.array(cc["ang_vel_variance"])
        e.vel_variance = sdiff * np.array(cc["vel_variance"])
        e.pendulum_rp_variance = sdiff * np.array(cc["pendulum_rp_variance"])
        e.pendulum_ang_vel_variance = sdiff * np.array(cc["pendulum_ang_vel_variance"])
        reps = 8
        zs, us, qps, qvs = [], [], [], []
        for r in range(reps):
            rec = RecordingRNG(1000 * ci + r)
            e._np_random = rec
            e.np_random = rec
            qp, qv = e.sample_state()
            z = np.zeros(15); u = np.zeros(2)
            z[:len(rec.z)] = rec.z
            u[:len(rec.u)] = rec.u
            zs.append(z); us.append(u); qps.append(np.asarray(qp, dtype=float)); qvs.append(np.asarray(qv, dtype=float))
        k = "ss%d_" % ci
        out[k + "cfg"] = np.array(json.dumps(cc))
        out[k + "z"], out[k + "u"] = np.array(zs), np.array(us)
        out[k + "qpos"], out[k + "qvel"] = np.array(qps), np.array(qvs)
    # anchor quoted in SURVEY.md 8c: default_rng(43), load, offset 0.8, vel/angvel 0.4, pend 0.2
    e = object.__new__(bde.BaseDroneEnv)
    e.pendulum, e.random_start_pos, e.start_pos, e.max_pos_offset = True, True, [0, 0, 15, 0], 0.8
    e.angle_variance = np.zeros(2)
    e.vel_variance = e.ang_vel_variance = 0.4 * np.ones(3)
    e.pendulum_rp_variance = e.pendulum_ang_vel_variance = 0.2 * np.ones(2)
    e._np_random = e.np_random = np.random.default_rng(43)
    qp, qv = e.sample_state()
    out["ss_anchor_qpos"], out["ss_anchor_qvel"] = np.asarray(qp), np.asarray(qv)

    # ------------------------------------------------------------------ a11
    for ci, (rp, pend, diff) in enumerate([(True, True, 1.0), (True, False, 0.1), (False, True, 1.0)]):
        e = object.__new__(bde.BaseDroneEnv)
        e.mass_interval = np.array(base["mass_interval"])
        e.arm_len_interval = np.array(base["arm_len_interval"])
        e.motor_force_interval = np.array(base["motor_force_interval"])
        e.motor_tau_interval = np.array(base["motor_tau_interval"])
        e.pendulum_length_interval = np.array(base["pendulum_length_interval"])
        e.weight_mass_interval = np.array(base["weight_mass_interval"])
        e.random_params, e.pendulum, e.param_difficulty, e.num_drones = rp, pend, diff, 7
        rec = RecordingRNG(77 + ci)
        e._np_random = e.np_random = rec
        params = e.generate_drone_params()
        k = "gp%d_" % ci
        out[k + "cfg"] = np.array(json.dumps(dict(random_params=rp, pendulum=pend, param_difficulty=diff)))
        out[k + "u"] = np.array(rec.u).reshape(6, 7) if rec.u else np.zeros((6, 7))
        out[k + "params"] = np.array([[float(v) for v in p.values()] for p in params])
    out["gp_centers"] = np.array([base[k][0] for k in ("mass_interval", "arm_len_interval", "motor_force_interval",
                                                        "motor_tau_interval", "pendulum_length_interval",
                                                        "weight_mass_interval")])
    out["gp_widths"] = np.array([base[k][1] for k in ("mass_interval", "arm_len_interval", "motor_force_interval",
                                                       "motor_tau_interval", "pendulum_length_interval",
                                                       "weight_mass_interval")])

    # ------------------------------------------------------------------ a13
    nd = 6
    qpos = rng.normal(size=nd * 7)
    for i in range(nd):
        q = qpos[7 * i + 3:7 * i + 7]
        qpos[7 * i + 3:7 * i + 7] = q / np.linalg.norm(q)
    qpos[3:7] += rng.uniform(-0.03, 0.03, 4)  # reset_model perturbs the quaternion unnormalised
    e = object.__new__(sd.SimpleDrone)
    e.num_drones = nd
    e.data = types.SimpleNamespace(qpos=qpos)
    e.reference = [0, 0, 1]
    e.frame_skip = 2
    e.do_simulation = lambda a, n: None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ob, rew, term, _ = e.step(np.ones(4 * nd) * 0.7)
    out["sd_qpos"], out["sd_obs"] = qpos, np.asarray(ob)
    out["sd_reward"], out["sd_terminated"] = np.float64(rew), np.bool_(term)

    # ------------------------------------------------------------------ 8f-3: analytic PID cascade
    from models.Analytic.AttitudeController import AttittudeController
    from models.Analytic.PositionController import PositionController
    nd, T = 6, 12
    masses = rng.uniform(1.3, 1.9, nd)
    forces = rng.uniform(6, 8, nd)
    attc, posc = AttittudeController(nd, masses, forces), PositionController(nd)
    pid_ref = np.array([0.4, -0.3, 10.0, 0.6])
    xyz_seq = pid_ref[:3, None, None] + rng.normal(scale=1.2, size=(3, T, nd))
    xyz_seq[:, 3] += 4.0                                   # exercises the +-2 error clip and the output clips
    rpy_seq = rng.uniform(-0.6, 0.6, size=(3, T, nd))
    rpy_seq[2] = rng.uniform(-np.pi, np.pi, size=(T, nd))
    pos_out, rpyz_out, ctrl_out = [], [], []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for t in range(T):
            pa = posc.compute_control(pid_ref[:3], xyz_seq[:, t])
            rz = attc.tilts2rpy(pa, np.ones(nd) * pid_ref[3])
            ct = attc.compute_control(rz, rpy_seq[:, t])
            pos_out.append(np.array(pa)); rpyz_out.append(np.array(rz)); ctrl_out.append(np.array(ct))
    out["pid_masses"], out["pid_forces"], out["pid_ref"] = masses, forces, pid_ref
    out["pid_xyz"], out["pid_rpy"] = xyz_seq, rpy_seq
    out["pid_pos_action"], out["pid_rpyz"], out["pid_ctrl"] = np.array(pos_out), np.array(rpyz_out), np.array(ctrl_out)

    # ------------------------------------------------------------------ 8f-4: trajectory generators
    # evaluation.py cannot be imported (ray, matplotlib, the policy models at module scope); its three pure
    # generator functions are compiled from the file where it lies and executed on their own.
    import ast
    with open(os.path.join(REF, "evaluation.py")) as fh:
        tree = ast.parse(fh.read())
    wanted = ("gen_circle_trajectory", "gen_step_trajectory", "gen_ramp_trajectory")
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted], type_ignores=[])
    ns = {"np": np}
    exec(compile(mod, os.path.join(REF, "evaluation.py"), "exec"), ns)
    out["traj_circle_args"] = np.array([2.0, 0.5, 1.0, 15.0])          # T, f, r, h
    out["traj_circle"] = ns["gen_circle_trajectory"](T=2.0, f=0.5, r=1.0, h=15.0)[1]
    out["traj_start"], out["traj_end"] = np.array([0.5, -0.5, 15.0, 0.0]), np.array([1.5, 0.25, 14.0, 0.6])
    out["traj_step_args"] = np.array([0.57, 1.5])                       # step_time, duration
    out["traj_step"] = ns["gen_step_trajectory"](0.57, 1.5, list(out["traj_start"]), list(out["traj_end"]))[1]
    out["traj_ramp_args"] = np.array([0.4, 1.6])                        # start_time, duration
    out["traj_ramp"] = ns["gen_ramp_trajectory"](0.4, 1.6, list(out["traj_start"]), list(out["traj_end"]))[1]
    out["traj_step_default"] = ns["gen_step_trajectory"]()[1]
    out["traj_ramp_default"] = ns["gen_ramp_trajectory"]()[1]

    # ------------------------------------------------------------------ a14
    cfg = {k: v for k, v in base.items() if not callable(v)}
    out["base_config_json"] = np.array(json.dumps(cfg))
    out["base_config_reward_fcn"] = np.array(base["reward_fcn"].__name__)
    out["base_config_terminated_fcn"] = np.array(base["terminated_fcn"].__name__)

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "keys:", len(out), "bytes:", os.path.getsize(OUT))


if __name__ == "__main__":
    main()
