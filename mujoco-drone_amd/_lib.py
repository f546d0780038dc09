"""ctypes binding of include/qd.h (the C ABI of libqd.so).

The library is the product: if it is missing or cannot be loaded this module raises --
there is no Python/CPU fallback for any of the calls below.
"""
import ctypes as C
import os

from . import build as _build
from .build import LIB

QD_OK, QD_ERR_INVALID, QD_ERR_SHAPE, QD_ERR_UNSUPPORTED, QD_ERR_HIP, QD_ERR_ARENA, QD_ERR_INDEX = 0, -1, -2, -3, -4, -5, -6
MODEL_NOLOAD, MODEL_LOAD = 0, 1
CTRL_DIRECT, CTRL_AFFINE = 0, 1
TERM_DEFAULT, TERM_SIMPLE = 0, 1
START_FIXED, START_RANDOM, START_SIMPLE = 0, 1, 2
REF_STATIC, REF_CIRCLE, REF_STEP, REF_RAMP = 0, 1, 2, 3
TF_QUAT2RPY, TF_RPY2QUAT, TF_QUAT2DCM, TF_DCM2QUAT, TF_PENDRP2QUAT = range(5)

# observation variants, in the order of include/qd.h (= file order of observation_wrappers.py)
OBS_KINDS = [
    "BaseDroneEnv", "GlobalFrameRPYEnv", "LocalFramePRYEnv", "LocalFrameFullStateEnv", "LocalFrameFullStateZvecEnv",
    "LocalFramePRYaccEnv", "LocalFramePRYParamsEnv", "LocalFramePRYaccParamsEnv", "LocalFrameRPYParamsEnv",
    "LocalFrameRPYFakeParamsEnv", "LocalFrameRPYEnv", "LocalFramePRYaccNoPendEnv", "LocalFramePRYaccParamsNoPendEnv",
    "LocalFrameRmParamsEnv", "LocalFrameZvecEnv", "SimpleDrone",
]
# reward functions, in the order of include/qd.h (= file order of rewards.py, then SimpleDrone.step)
REWARD_KINDS = [
    "default_reward_fcn", "distance_reward_fcn", "distance_energy_reward", "distance_energy_reward_pendulum_angle",
    "distance_energy_reward_pendulum_angle2", "distance_energy_reward_pendulum_angle3",
    "distance_energy_reward_pendulum_en", "distance_energy_reward_pendulum_en2", "distance_energy_reward_pendulum_en3",
    "distance_energy_reward_pendulum_en4", "distance_time_energy_reward", "reward_1", "reward_pendulum_dist",
    "reward_pendulumDistHeading", "reward_2", "reward_2_penergy", "reward_3", "simple_drone_reward",
]


class QdConfig(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32), ("model", C.c_int32), ("obs_kind", C.c_int32), ("reward_kind", C.c_int32),
        ("frame_skip", C.c_int32), ("max_steps", C.c_int32), ("ctrl_map", C.c_int32), ("term_kind", C.c_int32),
        ("random_start", C.c_int32), ("random_params", C.c_int32), ("auto_reset", C.c_int32),
        ("per_env_reference", C.c_int32),
        ("timestep", C.c_double), ("max_distance", C.c_double), ("reference", C.c_double * 4),
        ("start_pos", C.c_double * 4), ("max_pos_offset", C.c_double),
        ("angle_var", C.c_double * 2), ("vel_var", C.c_double * 3), ("ang_vel_var", C.c_double * 3),
        ("pend_rp_var", C.c_double * 2), ("pend_vel_var", C.c_double * 2),
        ("param_center", C.c_double * 6), ("param_width", C.c_double * 6), ("param_difficulty", C.c_double),
        ("seed", C.c_uint64),
        ("ref_mode", C.c_int32), ("floor_contact", C.c_int32), ("ref_radius", C.c_double), ("ref_frequency", C.c_double),
        ("ref_t0", C.c_double), ("ref_duration", C.c_double), ("ref_end", C.c_double * 4),
    ]


class QdPolicyOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("in_buf", C.c_int32), ("in_off", C.c_int32), ("in_dim", C.c_int32),
                ("out_buf", C.c_int32), ("out_off", C.c_int32), ("out_dim", C.c_int32), ("act", C.c_int32),
                ("flags", C.c_int32), ("reserved0", C.c_int32), ("w_off", C.c_int64), ("b_off", C.c_int64)]


class QdPolicyRing(C.Structure):
    _fields_ = [("rows", C.c_int32), ("width", C.c_int32), ("period", C.c_int32), ("reserved0", C.c_int32),
                ("fill_off", C.c_int64)]


class QdPolicyDesc(C.Structure):
    _fields_ = [("n_ops", C.c_int32), ("n_bufs", C.c_int32), ("buf_width", C.c_int32 * 8), ("obs_dim", C.c_int32),
                ("act_dim", C.c_int32), ("logits_buf", C.c_int32), ("logits_off", C.c_int32), ("n_logits", C.c_int32),
                ("value_buf", C.c_int32), ("value_off", C.c_int32), ("n_rings", C.c_int32), ("ring", QdPolicyRing * 4),
                ("aux_buf", C.c_int32), ("aux_off", C.c_int32), ("aux_dim", C.c_int32), ("dist", C.c_int32)]


POL_DENSE, POL_AFFINE, POL_COPY_OBS, POL_COPY_PREV, POL_RING_LOAD, POL_RING_PUSH, POL_LSTM_CELL = 0, 1, 2, 3, 4, 5, 6
ACT_NONE, ACT_TANH, ACT_RELU = 0, 1, 2
POL_VALUE_ONLY = 1
DIST_BETA, DIST_SQUASHED_GAUSSIAN = 0, 1
OPT_PERSISTENT_FRAGMENTS = 0
OPT_LATENCY_KERNEL = 1

# every symbol include/qd.h declares: (restype, argtypes)
_VP, _I, _I64 = C.c_void_p, C.c_int, C.c_int64
_D4 = C.POINTER(C.c_double)
SIGNATURES = {
    "qd_last_error": (C.c_char_p, []),
    "qd_version": (_I, []),
    "qd_source_hash": (C.c_char_p, []),
    "qd_obs_dim": (_I, [_I, _I]),
    "qd_state_dim": (_I, [_I]),
    "qd_arena_bytes": (C.c_size_t, [_I]),
    "qd_create": (_I, [C.POINTER(QdConfig), _VP, C.c_size_t, C.POINTER(_VP)]),
    "qd_destroy": (_I, [_VP]),
    "qd_init": (_I, [_VP, _VP]),
    "qd_reset_data": (_I, [_VP, _VP]),
    "qd_set_reference": (_I, [_VP, _D4]),
    "qd_set_reference_per_env": (_I, [_VP, _VP, _VP]),
    "qd_set_reference_schedule": (_I, [_VP, _D4, _I]),
    "qd_randomize_params": (_I, [_VP, _VP]),
    "qd_set_params": (_I, [_VP, _VP, _VP]),
    "qd_get_params": (_I, [_VP, _VP, _VP]),
    "qd_reset": (_I, [_VP, _VP, _VP, _VP]),
    "qd_reset_at": (_I, [_VP, _I, _VP]),
    "qd_set_state": (_I, [_VP, _VP, _VP, _VP, _VP]),
    "qd_get_state": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "qd_step": (_I, [_VP, _VP, _I64, _VP, _VP, _VP, _VP]),
    "qd_rollout": (_I, [_VP, _VP, _I, _VP, _VP, _VP, _VP]),
    "qd_step_fragment": (_I, [_VP, _VP, _I, _VP, _VP, _VP, _VP]),
    "qd_set_option": (_I, [_VP, _I, _I]),
    "qd_health_counters": (_I, [_VP, _VP, _VP]),
    "qd_step_kernel_name": (C.c_char_p, [_VP]),
    "qd_fragment_kernel_name": (C.c_char_p, [_VP]),
    "qd_pool_counters": (_I, [_VP, _VP, _VP]),
    "qd_pid_reset": (_I, [_VP, _VP, _VP]),
    "qd_pid_action": (_I, [_VP, _VP, _VP]),
    "qd_rollout_pid": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _VP]),
    "qd_policy_packed_bytes": (C.c_size_t, [C.POINTER(QdPolicyDesc), C.POINTER(QdPolicyOp)]),
    "qd_policy_create": (_I, [C.POINTER(QdPolicyDesc), C.POINTER(QdPolicyOp), _VP, C.c_size_t, _VP, C.c_size_t,
                              C.POINTER(_VP)]),
    "qd_policy_destroy": (_I, [_VP]),
    "qd_policy_kernel": (_I, [_VP]),
    "qd_policy_forward": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "qd_policy_act": (_I, [_VP, _I, _VP, _VP, _VP, _I, C.c_uint64, C.c_uint32, _VP, _VP, _VP, _VP, _VP, _VP]),
    "qd_rollout_policy": (_I, [_VP, _VP, _I, _VP, _VP, _I, C.c_uint64, C.c_uint32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "qd_policy_aux": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _VP]),
    "qd_policy_state_bytes": (C.c_size_t, [_VP, _I]),
    "qd_policy_reset_state": (_I, [_VP, _VP, _I, _VP, _VP]),
    "qd_observe": (_I, [_VP, _VP, _VP]),
    "qd_drone_states": (_I, [_VP, _VP, _VP]),
    "qd_eval_obs": (_I, [_I, _I, _VP, _D4, _VP, _I, _VP]),
    "qd_eval_reward": (_I, [_I, _I, _VP, _VP, _VP, _D4, C.c_double, _VP, _I, _VP]),
    "qd_eval_truncated": (_I, [_I, _VP, _VP, _D4, C.c_double, _I, _VP, _I, _VP]),
    "qd_transform": (_I, [_I, _VP, _VP, _I, _VP]),
    "qd_column_stats_workspace_bytes": (C.c_size_t, [_I]),
    "qd_column_stats": (_I, [_VP, _I64, _I, _VP, _VP, C.c_size_t, _VP]),
    "qd_episode_stats_workspace_bytes": (C.c_size_t, [_I]),
    "qd_episode_stats": (_I, [_VP, _VP, _I, _I, _VP, _VP, _VP, C.c_size_t, _VP]),
}

_lib = None


class QdError(RuntimeError):
    pass


def lib():
    """Load libqd.so (once).  Raises ImportError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise ImportError(
                "libqd.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`python -m mujoco_drone_amd.build`; there is no CPU fallback." % LIB)
        # torch first: its bundled HIP runtime must be the one already in the process when libqd.so
        # resolves libamdhip64 (loading the system copy first leaves two runtimes fighting for the device)
        import torch  # noqa: F401
        handle = C.CDLL(LIB)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        # a prebuilt library must come from exactly the sources beside it (QD_LIB = a diagnostic build, exempt)
        if not os.environ.get("QD_LIB") and not os.environ.get("QD_ALLOW_STALE_LIB"):
            have, want = handle.qd_source_hash().decode(), _build.source_hash(_build._extra_flags())
            if have != want:
                raise ImportError("libqd.so was built from other sources (library %s..., tree %s...): rebuild with "
                                  "`python -m mujoco_drone_amd.build` or __graft_entry__.build()" % (have[:12], want[:12]))
        _lib = handle
    return _lib


def last_error():
    return lib().qd_last_error().decode("utf-8", "replace")


def check(rc):
    """Translate a qd_status into the exception type the reference raises for the same condition."""
    if rc == QD_OK:
        return
    msg = last_error()
    if rc == QD_ERR_SHAPE:
        raise ValueError(msg)            # mujoco_env_custom.py:200-201
    if rc == QD_ERR_INDEX:
        raise AssertionError(msg)        # BaseDroneEnv.py:338
    if rc == QD_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc in (QD_ERR_INVALID, QD_ERR_ARENA):
        raise ValueError(msg)
    raise QdError("qd error %d: %s" % (rc, msg))


def double4(v):
    v = [float(x) for x in v]
    v = (v + [0.0, 0.0, 0.0, 0.0])[:4]
    return (C.c_double * 4)(*v)
