"""The reference's observation variants (environments/observation_wrappers.py:7-528) as
subclasses that only select which fused observation the step kernel emits.  Class names,
`num_states` / `num_params` and the layouts are the reference's; see DESIGN.md (and
csrc/qd_obsrew.h) for the layouts and the quirks that are reproduced:
  * LocalFrameFullStateZvecEnv declares 23 states but emits 24 values (:121,149): the
    observation_space here has the emitted length;
  * LocalFramePRYaccParamsNoPendEnv raises NameError('acc') on first use in the reference
    (:438,448); here the constructor raises it;
  * without the load the wrappers slice the 29-element state with the 33-element offsets
    (:404), so e.g. "acc" is act[1:4] and `params` has 2 entries.
"""
import numpy as np

from .BaseDroneEnv import BaseDroneEnv, Box
from .. import _lib as L


def _variant(name, num_states, num_params, doc):
    kind = L.OBS_KINDS.index(name)

    def __init__(self, config, **kwargs):
        BaseDroneEnv.__init__(self, config, **kwargs)
        self.num_states = num_states
        self.num_params = num_params
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(self._dev.D,), dtype=np.float64)

    return type(name, (BaseDroneEnv,), {"OBS_KIND": kind, "__init__": __init__, "__doc__": doc,
                                        "__module__": __name__})


GlobalFrameRPYEnv = _variant("GlobalFrameRPYEnv", 16, 0,
                             "e_g, roll, pitch, heading diff, world vel, body rates, pendulum rp, pendulum rates (:7-35)")
LocalFramePRYEnv = _variant("LocalFramePRYEnv", 16, 0,
                            "e_l, pitch, roll, heading diff, local vel, body rates, pendulum pr, pendulum rates (:38-73)")
LocalFrameFullStateEnv = _variant("LocalFrameFullStateEnv", 23, 0,
                                  "LocalFramePRYEnv + accelerometer + activations before the pendulum part (:76-111)")
LocalFrameFullStateZvecEnv = _variant("LocalFrameFullStateZvecEnv", 23, 0,
                                      "like LocalFrameFullStateEnv with the body z vector instead of pitch/roll; "
                                      "24 values (:114-151)")
LocalFramePRYaccEnv = _variant("LocalFramePRYaccEnv", 19, 0, "LocalFramePRYEnv + accelerometer (:154-191)")
LocalFramePRYParamsEnv = _variant("LocalFramePRYParamsEnv", 16, 6, "LocalFramePRYEnv + drone parameters (:194-230)")
LocalFramePRYaccParamsEnv = _variant("LocalFramePRYaccParamsEnv", 19, 6,
                                     "pendulum pr, acc, pendulum rates, parameters (:233-265)")
LocalFrameRPYParamsEnv = _variant("LocalFrameRPYParamsEnv", 16, 6,
                                  "e_l, roll, pitch, heading diff, local vel, body rates, pendulum rp, pendulum rates, "
                                  "parameters: the train_PPO / train_RMA observation (:268-304)")
LocalFrameRPYFakeParamsEnv = _variant("LocalFrameRPYFakeParamsEnv", 16, 6,
                                      "LocalFrameRPYParamsEnv with constant parameters [1,0.17,7,0.01,1.2,0.3] (:307-344)")
LocalFrameRPYEnv = _variant("LocalFrameRPYEnv", 16, 0, "LocalFrameRPYParamsEnv without parameters (:347-382)")
LocalFramePRYaccNoPendEnv = _variant("LocalFramePRYaccNoPendEnv", 15, 0, "local state + state[16:19] (:385-416)")
LocalFramePRYaccParamsNoPendEnv = _variant("LocalFramePRYaccParamsNoPendEnv", 15, 6,
                                           "broken in the reference: NameError('acc') (:419-450)")
LocalFrameRmParamsEnv = _variant("LocalFrameRmParamsEnv", 22, 6,
                                 "e_l, 3x3 rotation relative to the reference yaw, local vel, body rates, pendulum, "
                                 "parameters (:453-489)")
LocalFrameZvecEnv = _variant("LocalFrameZvecEnv", 17, 0, "e_l, body z vector, heading diff, ... (:492-528)")
