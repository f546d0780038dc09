"""Device-side env batch: owns the torch tensors (arena + output buffers) and the
qd_env handle, and exposes every C-ABI call as a method taking/returning torch
tensors.  PyTorch is used for device memory and streams only."""
import ctypes as C

import numpy as np
import torch

from .. import _lib as L


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


try:  # raw hipStream_t of torch's current stream without building a Stream object (hot path)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:  # pragma: no cover
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream


# float4 planes of the arena, in the order of `enum Group` in csrc/qd_kernels.hip (tests/test_host_logic.py keeps the two in step)
ARENA_PLANES = ["POS", "QUAT", "VEL", "ANG", "ACT", "AUX", "ACC", "M0", "M1", "M2", "M3", "M4", "M5", "M6", "P0", "P1", "REF",
                "NX0", "NX1", "NX2", "NX3", "NX4", "NY0", "NY1", "NY2", "NY3", "NY4", "NXA0", "NXA1", "NXA2", "NXA3",
                "NYA0", "NYA1", "NYA2", "NYA3", "C0", "C1", "C2", "C3"]


class DeviceEnv:
    def __init__(self, cfg: L.QdConfig, device="cuda:0"):
        self.lib = L.lib()  # raises if libqd.so is missing: no CPU path
        if not torch.cuda.is_available():
            raise RuntimeError("mujoco_drone_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.device = torch.device(device)
        self.cfg = cfg
        self.n = int(cfg.num_envs)
        self.load = cfg.model == L.MODEL_LOAD
        self.nq, self.nv = (9, 8) if self.load else (7, 6)
        self.ns = self.lib.qd_state_dim(cfg.model)
        self.D = self.lib.qd_obs_dim(cfg.obs_kind, cfg.model)
        nbytes = self.lib.qd_arena_bytes(self.n)
        with torch.cuda.device(self.device):
            # torch's caching allocator hands out 512-byte aligned blocks
            self.arena = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
            self.obs = torch.zeros((self.n, max(self.D, 1)), dtype=torch.float32, device=self.device)
            self.reward = torch.zeros(self.n, dtype=torch.float32, device=self.device)
            self.truncated = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
        handle = C.c_void_p()
        L.check(self.lib.qd_create(C.byref(cfg), _ptr(self.arena), nbytes, C.byref(handle)))
        self.handle = handle
        self._qd_step = self.lib.qd_step
        self._obs_ptr, self._rew_ptr, self._trunc_ptr = self.obs.data_ptr(), self.reward.data_ptr(), self.truncated.data_ptr()
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device(self.device.type, self._dev_index)   # canonical form: tensors report an indexed device
        self._frag_cache = {}
        self._qd_step_fragment = self.lib.qd_step_fragment
        L.check(self.lib.qd_init(self.handle, self._stream()))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.qd_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _f32(self, x, shape):
        t = torch.as_tensor(x)
        if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
        if tuple(t.shape) != tuple(shape):
            if t.numel() != int(np.prod(shape)):
                raise ValueError("expected shape %s, got %s" % (tuple(shape), tuple(t.shape)))
            t = t.reshape(shape)
        return t

    # ------------------------------------------------------------------ calls
    def set_reference(self, ref):
        L.check(self.lib.qd_set_reference(self.handle, L.double4(ref)))

    def set_reference_per_env(self, ref):
        ref = self._f32(ref, (self.n, 4))
        L.check(self.lib.qd_set_reference_per_env(self.handle, _ptr(ref), self._stream()))

    def randomize_params(self):
        L.check(self.lib.qd_randomize_params(self.handle, self._stream()))

    def set_params(self, raw):
        raw = torch.as_tensor(np.asarray(raw, dtype=np.float64)).to(self.device).contiguous().reshape(self.n, 6)
        L.check(self.lib.qd_set_params(self.handle, _ptr(raw), self._stream()))

    def get_params(self):
        out = torch.empty((self.n, 6), dtype=torch.float64, device=self.device)
        L.check(self.lib.qd_get_params(self.handle, _ptr(out), self._stream()))
        return out

    def reset_data(self):
        L.check(self.lib.qd_reset_data(self.handle, self._stream()))

    def reset(self, mask=None, want_obs=True):
        if mask is not None:
            mask = torch.as_tensor(mask).to(device=self.device, dtype=torch.uint8).contiguous()
            if mask.numel() != self.n:
                raise ValueError("mask must have one entry per env")
        L.check(self.lib.qd_reset(self.handle, _ptr(mask), _ptr(self.obs) if want_obs else None, self._stream()))
        return self.obs if want_obs else None

    def reset_at(self, index):
        L.check(self.lib.qd_reset_at(self.handle, int(index), self._stream()))

    def set_state(self, qpos, qvel, act=None):
        qpos = self._f32(qpos, (self.n, self.nq))
        qvel = self._f32(qvel, (self.n, self.nv))
        act = self._f32(act, (self.n, 4)) if act is not None else None
        L.check(self.lib.qd_set_state(self.handle, _ptr(qpos), _ptr(qvel), _ptr(act), self._stream()))

    def get_state(self):
        kw = dict(dtype=torch.float32, device=self.device)
        qpos, qvel = torch.empty((self.n, self.nq), **kw), torch.empty((self.n, self.nv), **kw)
        act, sens = torch.empty((self.n, 4), **kw), torch.empty((self.n, 3), **kw)
        steps = torch.empty(self.n, dtype=torch.int32, device=self.device)
        L.check(self.lib.qd_get_state(self.handle, _ptr(qpos), _ptr(qvel), _ptr(act), _ptr(sens), _ptr(steps),
                                      self._stream()))
        return qpos, qvel, act, sens, steps

    def step(self, actions, obs=None, reward=None, truncated=None):
        """actions: float32 device tensor with 4*N values (anything else raises ValueError like the reference).
        Hot path: one ctypes call -> one kernel launch on torch's current stream, no allocation, no sync."""
        if not (type(actions) is torch.Tensor and actions.dtype is torch.float32 and actions.is_cuda
                and actions.is_contiguous()) or actions.device != self.device:
            actions = torch.as_tensor(np.asarray(actions, dtype=np.float32) if not isinstance(actions, torch.Tensor)
                                      else actions).to(device=self.device, dtype=torch.float32).contiguous()
        if obs is None:
            rc = self._qd_step(self.handle, actions.data_ptr(), actions.numel(), self._obs_ptr, self._rew_ptr,
                               self._trunc_ptr, _raw_stream(self._dev_index))
            if rc:
                L.check(rc)
            return self.obs, self.reward, self.truncated
        rc = self._qd_step(self.handle, actions.data_ptr(), actions.numel(), obs.data_ptr(), reward.data_ptr(),
                           truncated.data_ptr(), _raw_stream(self._dev_index))
        if rc:
            L.check(rc)
        return obs, reward, truncated

    def step_fragment(self, actions, obs, reward, truncated):
        """T env steps by one C call: actions [T,N,4] -> obs [T,N,D], reward [T,N], truncated [T,N], all caller-owned device
        tensors reused from call to call (one persistent launch, or T per-step launches replayed from a HIP graph captured on
        first use for these buffers: qd_step_fragment).

        The kernels write through raw pointers: a strided view, another dtype or another device would be written out of bounds or
        as garbage, so every buffer is validated -- once per set of tensor OBJECTS.  The cache is keyed by the four objects'
        identities and holds references to them: while an entry lives its ids cannot be handed to other tensors (a key made of
        data_ptr()s could: the caching allocator recycles addresses, and a recycled address says nothing about dtype or strides).
        A hit costs four id()s, a dict lookup, four data_ptr()s (an in-place set_ / resize_ of a cached tensor is caught by them)
        and the ctypes call."""
        ent = self._frag_cache.get((id(actions), id(obs), id(reward), id(truncated)))
        if ent is not None:
            ap, op, rp, tp = actions.data_ptr(), obs.data_ptr(), reward.data_ptr(), truncated.data_ptr()
            if (ap, op, rp, tp) == ent[1]:
                rc = self._qd_step_fragment(self.handle, ap, ent[0], op, rp, tp, _raw_stream(self._dev_index))
                if rc:
                    L.check(rc)
                return obs, reward, truncated
        T = int(actions.shape[0])
        if tuple(actions.shape[1:]) != (self.n, 4) or actions.dtype != torch.float32 or not actions.is_contiguous():
            raise ValueError("Action dimension mismatch")
        if tuple(obs.shape) != (T, self.n, self.D) or tuple(reward.shape) != (T, self.n) or tuple(truncated.shape) != (T, self.n):
            raise ValueError("fragment buffers must be [T,N,D], [T,N], [T,N]")
        for name, t, dt in (("actions", actions, torch.float32), ("obs", obs, torch.float32), ("reward", reward, torch.float32),
                            ("truncated", truncated, torch.uint8)):
            if t.dtype != dt or not t.is_contiguous() or t.device != self.device:
                raise ValueError("fragment buffer %s must be a contiguous %s tensor on %s (got %s, contiguous=%s, %s)"
                                 % (name, dt, self.device, t.dtype, t.is_contiguous(), t.device))
        ptrs = (actions.data_ptr(), obs.data_ptr(), reward.data_ptr(), truncated.data_ptr())
        if len(self._frag_cache) >= 64:      # a sampler alternates between a few sets of buffers; a caller that never repeats one
            self._frag_cache.clear()         # does not grow the cache (nor keep its tensors alive) without bound
        self._frag_cache[(id(actions), id(obs), id(reward), id(truncated))] = (T, ptrs, (actions, obs, reward, truncated))
        L.check(self._qd_step_fragment(self.handle, ptrs[0], T, ptrs[1], ptrs[2], ptrs[3], _raw_stream(self._dev_index)))
        return obs, reward, truncated

    def set_option(self, option, value):
        """launch-variant switches (qd_set_option), e.g. set_option(L.OPT_PERSISTENT_FRAGMENTS, 0)"""
        L.check(self.lib.qd_set_option(self.handle, int(option), int(value)))

    def step_kernel_name(self):
        return self.lib.qd_step_kernel_name(self.handle).decode()

    def fragment_kernel_name(self):
        return self.lib.qd_fragment_kernel_name(self.handle).decode()

    def pool_counters(self):
        """(in-kernel resets served by the reset pool, in-kernel resets sampled inline) since construction"""
        out = torch.zeros(2, dtype=torch.int32, device=self.device)
        L.check(self.lib.qd_pool_counters(self.handle, _ptr(out), self._stream()))
        taken, inline = (int(x) & 0xFFFFFFFF for x in out.cpu().tolist())
        return taken, inline

    def health_counters(self):
        """events that must not happen since construction (qd_health_counters): (in-kernel polls that ran out,)"""
        out = torch.zeros(1, dtype=torch.int32, device=self.device)
        L.check(self.lib.qd_health_counters(self.handle, _ptr(out), self._stream()))
        return tuple(int(x) & 0xFFFFFFFF for x in out.cpu().tolist())

    def rollout(self, actions, obs=None, reward=None, truncated=None):
        """actions [T,N,4] -> obs [T,N,D], reward [T,N], truncated [T,N] in one launch."""
        actions = self._f32(actions, tuple(actions.shape))
        T = int(actions.shape[0])
        if tuple(actions.shape[1:]) != (self.n, 4):
            raise ValueError("Action dimension mismatch")
        kw = dict(device=self.device)
        obs = torch.empty((T, self.n, self.D), dtype=torch.float32, **kw) if obs is None else obs
        reward = torch.empty((T, self.n), dtype=torch.float32, **kw) if reward is None else reward
        truncated = torch.empty((T, self.n), dtype=torch.uint8, **kw) if truncated is None else truncated
        L.check(self.lib.qd_rollout(self.handle, _ptr(actions), T, _ptr(obs), _ptr(reward), _ptr(truncated),
                                    self._stream()))
        return obs, reward, truncated

    # ---- the analytic PID cascade as an on-device action source (models/Analytic/*.py, attitude_test.py:26-47)
    def pid_reset(self, mask=None):
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            if mask.numel() != self.n:
                raise ValueError("mask must have one entry per env")
        L.check(self.lib.qd_pid_reset(self.handle, _ptr(mask) if mask is not None else None, self._stream()))

    def pid_action(self, out=None):
        out = torch.empty((self.n, 4), dtype=torch.float32, device=self.device) if out is None else out
        L.check(self.lib.qd_pid_action(self.handle, _ptr(out), self._stream()))
        return out

    def rollout_pid(self, T, want_actions=False):
        T = int(T)
        kw = dict(device=self.device)
        obs = torch.empty((T, self.n, self.D), dtype=torch.float32, **kw)
        reward = torch.empty((T, self.n), dtype=torch.float32, **kw)
        truncated = torch.empty((T, self.n), dtype=torch.uint8, **kw)
        # where the loop runs launch by launch (floor contact; the load model outside the persistent kernel) the action buffer is
        # what carries the actions from the controller launch to the step launch: always handed over
        actions = torch.empty((T, self.n, 4), dtype=torch.float32, **kw)
        L.check(self.lib.qd_rollout_pid(self.handle, T, _ptr(obs), _ptr(reward), _ptr(truncated), _ptr(actions), self._stream()))
        return (obs, reward, truncated, actions) if want_actions else (obs, reward, truncated)

    def observe(self, out=None):
        out = self.obs if out is None else out
        L.check(self.lib.qd_observe(self.handle, _ptr(out), self._stream()))
        return out

    def drone_states(self):
        out = torch.empty((self.n, self.ns), dtype=torch.float32, device=self.device)
        L.check(self.lib.qd_drone_states(self.handle, _ptr(out), self._stream()))
        return out

    def planes(self):
        """the arena's float4 planes as a [len(ARENA_PLANES), N, 4] float32 view (diagnostics and tests)"""
        npad = (self.n + 255) // 256 * 256
        ngroups = len(ARENA_PLANES)
        return self.arena[:ngroups * npad * 16].view(torch.float32).view(ngroups, npad, 4)[:, :self.n]

    def model_constants(self):
        """per-env derived model constants (qd_model.h), read straight from the arena planes"""
        g, first = self.planes(), ARENA_PLANES.index("M0")
        names = ["m0", "c0z", "I0x", "I0y", "I0z", "rot", "gearF", "gearT", "inv_tau", "m2", "lc", "I2t", "I2a", "klin0",
                 "kang0", "qlx0", "qly0", "qlz0", "qax0", "qay0", "qaz0", "klin2", "kang2", "qlt2", "qla2", "qat2", "qaa2"]
        flat = torch.cat([g[first + k] for k in range(7)], dim=1)
        return {k: flat[:, i].clone() for i, k in enumerate(names)}


# ---------------------------------------------------------------- stateless helpers
def _dev(device=None):
    if not torch.cuda.is_available():
        raise RuntimeError("mujoco_drone_amd needs a ROCm GPU; there is no CPU fallback")
    return torch.device(device or "cuda:0")


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def eval_obs(kind, states, ref, device=None):
    dev = _dev(device)
    st = torch.as_tensor(np.asarray(states, dtype=np.float32)).to(dev).contiguous()
    if st.dim() == 1:
        st = st[None]
    n, ns = st.shape
    lib = L.lib()
    D = lib.qd_obs_dim(kind, L.MODEL_LOAD if ns == 33 else L.MODEL_NOLOAD)
    if D < 0:
        raise NameError("name 'acc' is not defined")  # observation_wrappers.py:448
    out = torch.empty((n, D), dtype=torch.float32, device=dev)
    L.check(lib.qd_eval_obs(kind, ns, _ptr(st), L.double4(ref), _ptr(out), n, _stream(dev)))
    return out


def eval_reward(kind, states, actions, num_steps, ref, max_distance, device=None):
    dev = _dev(device)
    st = torch.as_tensor(np.asarray(states, dtype=np.float32)).to(dev).contiguous()
    if st.dim() == 1:
        st = st[None]
    n, ns = st.shape
    ac = torch.as_tensor(np.asarray(actions, dtype=np.float32)).to(dev).contiguous().reshape(n, 4)
    ks = torch.as_tensor(np.asarray(num_steps, dtype=np.int32)).to(dev).contiguous().reshape(n)
    out = torch.empty(n, dtype=torch.float32, device=dev)
    lib = L.lib()
    rc = lib.qd_eval_reward(kind, ns, _ptr(st), _ptr(ac), _ptr(ks), L.double4(ref), float(max_distance), _ptr(out), n,
                            _stream(dev))
    if rc == L.QD_ERR_UNSUPPORTED:
        raise IndexError(L.last_error())  # params[4] on the 29-vector raises IndexError in the reference
    L.check(rc)
    return out


def eval_truncated(states, num_steps, ref, max_distance, max_steps, device=None):
    dev = _dev(device)
    st = torch.as_tensor(np.asarray(states, dtype=np.float32)).to(dev).contiguous()
    if st.dim() == 1:
        st = st[None]
    n, ns = st.shape
    ks = torch.as_tensor(np.asarray(num_steps, dtype=np.int32)).to(dev).contiguous().reshape(n)
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    L.check(L.lib().qd_eval_truncated(ns, _ptr(st), _ptr(ks), L.double4(ref), float(max_distance), int(max_steps),
                                      _ptr(out), n, _stream(dev)))
    return out


def transform(which, x, in_dim, out_dim, device=None):
    dev = _dev(device)
    arr = np.asarray(x, dtype=np.float32)
    single = arr.ndim == 1 or (which == L.TF_DCM2QUAT and arr.shape == (3, 3))
    t = torch.as_tensor(arr.reshape(-1, in_dim)).to(dev).contiguous()
    out = torch.empty((t.shape[0], out_dim), dtype=torch.float32, device=dev)
    L.check(L.lib().qd_transform(which, _ptr(t), _ptr(out), t.shape[0], _stream(dev)))
    res = out.cpu().numpy().astype(np.float64)
    return res[0] if single else res
