"""The reference's reward functions (environments/rewards.py:5-368) as selectable
device kernels.  Each name below is an object that (a) can be put in
config['reward_fcn'] exactly like the reference's function and selects the fused
reward inside the step kernel, and (b) can be called with the reference signature
`(env, state, action, num_steps) -> float`, in which case the same device code is
evaluated on that one state (qd_eval_reward).  Formulae and the reference quirks they
reproduce are listed in DESIGN.md."""
import numpy as np

from .. import _lib as L


class RewardFcn:
    def __init__(self, name, kind, doc):
        self.__name__ = name
        self.__qualname__ = name
        self.kind = kind
        self.__doc__ = doc

    def __call__(self, env, state, action, num_steps):
        from ._device import eval_reward
        ref = np.asarray(env.reference, dtype=np.float64)
        out = eval_reward(self.kind, np.asarray(state), np.asarray(action), [int(num_steps)], ref,
                          getattr(env, "max_distance", 0.0))
        return float(out[0].item())

    def __repr__(self):
        return "<device reward %s (kind %d)>" % (self.__name__, self.kind)


def resolve(fcn):
    """config['reward_fcn'] -> kind.  Accepts our RewardFcn objects, a name, or any callable whose
    __name__ is one of the reference's reward functions (e.g. the reference's own rewards.py objects)."""
    if isinstance(fcn, RewardFcn):
        return fcn.kind
    name = fcn if isinstance(fcn, str) else getattr(fcn, "__name__", None)
    if name in L.REWARD_KINDS:
        return L.REWARD_KINDS.index(name)
    raise TypeError("reward_fcn %r is not one of the reward functions the device kernels implement (%s); "
                    "arbitrary Python callables cannot run inside the GPU step" % (fcn, ", ".join(L.REWARD_KINDS[:-1])))


_DOCS = {
    "default_reward_fcn": "3 - |pos - ref|  (rewards.py:5-10)",
    "distance_reward_fcn": "5 - |pos - ref| - 0.1 h  (rewards.py:13-20)",
    "distance_energy_reward": "3.5 - |pos - ref|^2 - 0.1 h - 0.2 |a|^2  (rewards.py:23-31)",
    "distance_energy_reward_pendulum_angle": "rewards.py:34-43",
    "distance_energy_reward_pendulum_angle2": "rewards.py:46-56",
    "distance_energy_reward_pendulum_angle3": "rewards.py:59-72",
    "distance_energy_reward_pendulum_en": "rewards.py:75-107",
    "distance_energy_reward_pendulum_en2": "rewards.py:110-146",
    "distance_energy_reward_pendulum_en3": "rewards.py:149-188",
    "distance_energy_reward_pendulum_en4": "rewards.py:191-230",
    "distance_time_energy_reward": "rewards.py:233-242",
    "reward_1": "rewards.py:245-257",
    "reward_pendulum_dist": "rewards.py:283-294",
    "reward_pendulumDistHeading": "rewards.py:297-310",
    "reward_2": "rewards.py:313-327",
    "reward_2_penergy": "rewards.py:330-348",
    "reward_3": "rewards.py:351-368",
    "simple_drone_reward": "0.1 - |pos - ref|  (SimpleDrone.py:60)",
}
for _k, _name in enumerate(L.REWARD_KINDS):
    globals()[_name] = RewardFcn(_name, _k, _DOCS[_name])
del _k, _name
