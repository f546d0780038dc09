"""CPU-side checks of the C ABI: the library loads, exports every symbol include/qd.h declares, answers the
pure host queries and validates configurations -- no kernel is launched (there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    import importlib
    bld = importlib.import_module("mujoco_drone_amd.build")
    bld.build_library()
    from mujoco_drone_amd import _lib
    _lib.lib()
    return _lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "qd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qd_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(L):
    names = declared_symbols()
    assert len(names) >= 25
    raw = C.CDLL(L.LIB)
    for n in names:
        assert hasattr(raw, n), "libqd.so does not export %s" % n
    assert sorted(L.SIGNATURES) == names, "ctypes binding and include/qd.h disagree"


def test_enums_in_header_match_python_tables(L):
    text = open(os.path.join(ROOT, "include", "qd.h")).read()
    obs = re.search(r"QD_OBS_RAW = 0,(.*?)QD_OBS_COUNT", text, re.S).group(1)
    assert len(re.findall(r"QD_OBS_[A-Z_0-9]+", obs)) + 1 == len(L.OBS_KINDS) == 16
    rew = re.search(r"QD_REW_DEFAULT = 0,(.*?)QD_REW_COUNT", text, re.S).group(1)
    assert len(re.findall(r"QD_REW_[A-Z_0-9]+", rew)) + 1 == len(L.REWARD_KINDS) == 18


def test_dimensions(L):
    lib = L.lib()
    assert lib.qd_version() == 3
    assert lib.qd_state_dim(L.MODEL_LOAD) == 33 and lib.qd_state_dim(L.MODEL_NOLOAD) == 29
    want = {"BaseDroneEnv": 33, "GlobalFrameRPYEnv": 16, "LocalFramePRYEnv": 16, "LocalFrameFullStateEnv": 23,
            "LocalFrameFullStateZvecEnv": 24, "LocalFramePRYaccEnv": 19, "LocalFramePRYParamsEnv": 22,
            "LocalFramePRYaccParamsEnv": 25, "LocalFrameRPYParamsEnv": 22, "LocalFrameRPYFakeParamsEnv": 22,
            "LocalFrameRPYEnv": 16, "LocalFramePRYaccNoPendEnv": 15, "LocalFramePRYaccParamsNoPendEnv": -1,
            "LocalFrameRmParamsEnv": 28, "LocalFrameZvecEnv": 17, "SimpleDrone": 6}
    for name, d in want.items():
        assert lib.qd_obs_dim(L.OBS_KINDS.index(name), L.MODEL_LOAD) == d, name
    # without the load `params = state[27:]` has two entries (reference quirk)
    assert lib.qd_obs_dim(L.OBS_KINDS.index("LocalFrameRPYParamsEnv"), L.MODEL_NOLOAD) == 18
    assert lib.qd_obs_dim(L.OBS_KINDS.index("BaseDroneEnv"), L.MODEL_NOLOAD) == 29
    assert lib.qd_obs_dim(99, 1) == -1
    assert lib.qd_arena_bytes(0) == 0
    assert lib.qd_arena_bytes(4096) == 4096 * (39 * 16 + 19 * 8) + (4096 // 64 + 64) * 4   # float4 planes + float64 planes (6 raw parameters, 13 floor constants) + one refill counter per 64 envs + 64 words of statistics
    assert lib.qd_arena_bytes(4097) == 4352 * (39 * 16 + 19 * 8) + (4352 // 64 + 64) * 4   # padded to 256 envs


def _cfg(L, **kw):
    c = L.QdConfig()
    c.num_envs, c.model, c.obs_kind, c.reward_kind = 64, 1, 8, 2
    c.frame_skip, c.max_steps, c.ctrl_map, c.term_kind = 1, 512, 1, 0
    c.timestep, c.max_distance = 0.01, 4.0
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def test_create_validates_configuration_without_touching_the_gpu(L):
    lib = L.lib()
    arena = (C.c_char * (lib.qd_arena_bytes(64) + 256))()
    base = (C.addressof(arena) + 255) // 256 * 256
    h = C.c_void_p()

    def create(c, ptr=base, nbytes=None):
        return lib.qd_create(C.byref(c), C.c_void_p(ptr), nbytes if nbytes is not None else lib.qd_arena_bytes(64), C.byref(h))

    assert create(_cfg(L)) == 0 and h.value
    assert lib.qd_destroy(h) == 0
    assert create(_cfg(L, num_envs=0)) == L.QD_ERR_INVALID and b"num_envs" in lib.qd_last_error()
    assert create(_cfg(L, obs_kind=12)) == L.QD_ERR_UNSUPPORTED and b"NameError" in lib.qd_last_error()
    assert create(_cfg(L, obs_kind=77)) == L.QD_ERR_INVALID
    assert create(_cfg(L, model=0, reward_kind=9)) == L.QD_ERR_UNSUPPORTED and b"IndexError" in lib.qd_last_error()
    assert create(_cfg(L, obs_kind=15)) == L.QD_ERR_UNSUPPORTED          # SimpleDrone obs needs the no-load model
    assert create(_cfg(L, model=0, obs_kind=15, reward_kind=17)) == L.QD_ERR_INVALID   # ... and its termination rule
    assert create(_cfg(L, model=0, obs_kind=15, reward_kind=17, term_kind=1)) == 0
    lib.qd_destroy(h)
    assert create(_cfg(L, frame_skip=0)) == L.QD_ERR_INVALID
    assert create(_cfg(L, timestep=0.0)) == L.QD_ERR_INVALID
    assert create(_cfg(L), ptr=base + 4) == L.QD_ERR_ARENA and b"aligned" in lib.qd_last_error()
    assert create(_cfg(L), nbytes=1000) == L.QD_ERR_ARENA
    assert create(_cfg(L), ptr=0) == L.QD_ERR_ARENA
    # argument checks that come before any launch
    assert create(_cfg(L)) == 0
    assert lib.qd_step(h, None, 4 * 63, None, None, None, None) == L.QD_ERR_SHAPE
    assert lib.qd_last_error() == b"Action dimension mismatch"            # mujoco_env_custom.py:200-201
    assert lib.qd_step(h, None, 4 * 64, None, None, None, None) == L.QD_ERR_INVALID
    assert lib.qd_reset_at(h, 64, None) == L.QD_ERR_INDEX and lib.qd_reset_at(h, -1, None) == L.QD_ERR_INDEX
    assert lib.qd_set_reference_per_env(h, None, None) == L.QD_ERR_INVALID
    assert lib.qd_set_reference(h, L.double4([1, 2, 3, 0.5])) == 0
    lib.qd_destroy(h)
    assert lib.qd_step(None, None, 0, None, None, None, None) == L.QD_ERR_INVALID
    assert lib.qd_eval_obs(3, 31, None, L.double4([0] * 4), None, 4, None) == L.QD_ERR_INVALID
    assert lib.qd_eval_reward(14, 29, None, None, None, L.double4([0] * 4), 4.0, None, 4, None) == L.QD_ERR_UNSUPPORTED
    assert lib.qd_eval_obs(3, 33, None, L.double4([0] * 4), None, 0, None) == 0   # empty input is a no-op
    assert lib.qd_transform(9, None, None, 1, None) == L.QD_ERR_INVALID


def test_error_translation(L):
    with pytest.raises(ValueError):
        L.check(L.QD_ERR_SHAPE)
    with pytest.raises(AssertionError):
        L.check(L.QD_ERR_INDEX)
    with pytest.raises(NotImplementedError):
        L.check(L.QD_ERR_UNSUPPORTED)
    with pytest.raises(L.QdError):
        L.check(L.QD_ERR_HIP)
    L.check(0)


def test_header_is_plain_c_and_the_c_example_links(tmp_path):
    """include/qd.h must be consumable from C (the boundary is a C ABI): strict C99 syntax check of the header on its own, then the
    plain-C example is compiled and linked against libqd.so (it is RUN by the GPU suite, tests/test_gpu_capi_example.py)"""
    import subprocess
    probe = tmp_path / "probe.c"
    probe.write_text('#include "qd.h"\nint main(void) { qd_config c; (void)c; return qd_version() == QD_VERSION ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(probe)])
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("no HIP headers on this machine")
    libdir = os.path.join(ROOT, "mujoco-drone_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(rocm, "include"),
                           os.path.join(ROOT, "examples", "capi_hover.c"), "-o", str(tmp_path / "capi_hover"), "-L", libdir, "-lqd",
                           "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.join(rocm, "lib")])


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md maps each function of include/qd.h to the reference interface it replaces: none may be missing"""
    header = open(os.path.join(ROOT, "include", "qd.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    names = sorted(set(re.findall(r"\b(qd_[a-z_]+)\s*\(", header)))
    assert len(names) >= 40
    assert [n for n in names if n not in doc] == []
