"""GPU-backed `SimpleDrone` (environments/SimpleDrone.py:10-104): plain gym-style env with
`num_drones` load-free drones flattened into one observation, old 4-tuple `step`.

Reproduced reference behaviour: the action is applied directly as ctrl (no 0.1+0.9a,
:55); two physics substeps per step at make_sim's default 1000 Hz (:42-46); the
observation feeds MuJoCo's (w,x,y,z) quaternion to scipy as (x,y,z,w) and reads extrinsic
'zyx' angles (:95); reward and `terminated` look at drone 0 only (:57-60); reset adds
U(+-0.03) to every qpos coordinate including the quaternion and moves only drone 0 to
start_pos (:63-72).  Not reproduced: the render_fps assertion that fires in the reference
as written (mujoco_env_custom.py:122-124), rendering, and floor contact.
"""
import types

import numpy as np

from .. import _lib as L
from ._device import DeviceEnv
from .BaseDroneEnv import Box
from .env_gen import DEFAULT_FREQUENCY, DEFAULT_PARAMS, PARAM_NAMES


class SimpleDrone:
    metadata = {"render_modes": ["human", "rgb_array", "depth_array"], "render_fps": 50}

    def __init__(self, num_drones=1, reference=[0, 0, 1], start_pos=None, pendulum=False, **kwargs):
        if pendulum:
            raise NotImplementedError("the reference ignores `pendulum` here too: make_sim([{}]*n) has no load")
        self.num_drones = num_drones
        self.window_title = "test"
        self.reference = reference
        self.start_pos = self.reference[:3] if start_pos is None else start_pos
        self.render_mode = kwargs.get("render_mode", None)
        self.frame_skip = 2
        self.frequency = kwargs.get("frequency", DEFAULT_FREQUENCY)
        self.device = kwargs.get("device", "cuda:0")
        self.seed_value = int(kwargs.get("seed", 0))
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(self.num_drones * 6,), dtype=np.float64)
        self.action_space = Box(low=0.5, high=1, shape=(self.num_drones * 4,), dtype=np.float64)
        c = L.QdConfig()
        c.num_envs, c.model = int(num_drones), L.MODEL_NOLOAD
        c.obs_kind, c.reward_kind = L.OBS_KINDS.index("SimpleDrone"), L.REWARD_KINDS.index("simple_drone_reward")
        c.frame_skip, c.max_steps = self.frame_skip, 2 ** 31 - 1
        c.ctrl_map, c.term_kind = L.CTRL_DIRECT, L.TERM_SIMPLE
        c.random_start = int(kwargs.get("random_start", L.START_SIMPLE))
        c.random_params, c.auto_reset, c.per_env_reference = 0, int(bool(kwargs.get("auto_reset", False))), 0
        c.floor_contact = int(bool(kwargs.get("floor_contact", False)))   # extension: the floor of env_gen.py:97
        c.timestep = 1.0 / self.frequency
        c.max_distance = 0.5
        c.reference[:] = [float(x) for x in (list(self.reference) + [0.0])[:4]]
        c.start_pos[:] = [float(x) for x in (list(self.start_pos) + [0.0, 0.0])[:4]]
        c.param_center[:] = [DEFAULT_PARAMS[k] for k in PARAM_NAMES]  # make_drone defaults (env_gen.py:26-32)
        c.param_width[:] = [0.0] * 6
        c.param_difficulty = 0.0
        c.seed = self.seed_value
        self._dev = DeviceEnv(c, self.device)
        self.model = types.SimpleNamespace(nq=7 * num_drones, nv=6 * num_drones, nu=4 * num_drones,
                                           opt=types.SimpleNamespace(timestep=1.0 / self.frequency))
        self.terminated = np.zeros((self.num_drones,))

    @property
    def dt(self):
        return self.model.opt.timestep * self.frame_skip

    @property
    def data(self):
        qpos, qvel, act, sens, _ = self._dev.get_state()
        f = lambda t: t.cpu().numpy().astype(np.float64).ravel()
        return types.SimpleNamespace(qpos=f(qpos), qvel=f(qvel), act=f(act), sensordata=f(sens))

    def _get_obs(self):
        return self._dev.observe().cpu().numpy().astype(np.float64).ravel()

    def step(self, a):
        """SimpleDrone.py:54-61 -> (ob, reward, terminated, {})"""
        a = np.asarray(a, dtype=np.float32)
        if a.shape != (4 * self.num_drones,):
            raise ValueError("Action dimension mismatch")
        self._dev.set_reference((list(self.reference) + [0.0])[:4])
        obs, rew, term = self._dev.step(a)
        ob = obs.cpu().numpy().astype(np.float64).ravel()
        return ob, float(rew[0].item()), bool(term[0].item()), {}

    def step_tensor(self, actions, out=None):
        """zero-copy variant: actions [N,4] float32 CUDA tensor -> (obs [N,6], reward [N], terminated [N])"""
        o, r, t = out if out is not None else (None, None, None)
        return self._dev.step(actions, o, r, t)

    def step_fragment_tensor(self, actions, obs, reward, terminated):
        """T steps (actions [T,N,4] on the device) written in place into obs [T,N,6], reward [T,N], terminated [T,N]: the per-step
        kernels replayed from a HIP graph (qd_step_fragment)"""
        return self._dev.step_fragment(actions, obs, reward, terminated)

    def reset_model(self):
        self._dev.reset(None, want_obs=True)
        return self._dev.obs.cpu().numpy().astype(np.float64).ravel()

    def reset(self, *, seed=None, options=None):
        """SimpleDrone.py:74-79: mj_resetData + reset_model, returns the observation only"""
        self._dev.reset_data()
        return self.reset_model()

    def set_state(self, qpos, qvel):
        qpos, qvel = np.asarray(qpos, dtype=np.float64), np.asarray(qvel, dtype=np.float64)
        assert qpos.shape == (self.model.nq,) and qvel.shape == (self.model.nv,)
        self._dev.set_state(qpos.reshape(self.num_drones, 7), qvel.reshape(self.num_drones, 6))

    def render(self):
        return None

    def close(self):
        return None

    def viewer_setup(self):
        """camera placement of the reference's viewer (rendering is out of scope): accepted and ignored"""
        return None
