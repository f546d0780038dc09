"""Model generator.  The reference builds an MJCF document per drone with dm_control and
compiles it with MuJoCo (environments/env_gen.py:7-133).  Here the masses, centre of
mass, inertias, rotor geometry and gears are derived in closed form from the six drone
parameters, on the GPU, by qd_set_params / qd_randomize_params (csrc/qd_model.h); there
is no XML and no compile step.  This module keeps the parameter defaults of
make_drone and explains where the rest went."""

# make_drone defaults (env_gen.py:26-32)
DEFAULT_PARAMS = {"mass": 1.35, "arm_len": 0.15, "motor_force": 7.5, "motor_tau": 0.015, "pendulum_len": 0.0,
                  "weight_mass": 0.0}
PARAM_NAMES = ("mass", "arm_len", "motor_force", "motor_tau", "pendulum_len", "weight_mass")
DEFAULT_FREQUENCY = 1000  # make_sim(frequency=1000) (env_gen.py:76)


def params_to_row(params):
    """dict as in env.drone_params (BaseDroneEnv.py:208-214) -> [mass, arm_len, motor_force, motor_tau,
    pendulum_len, weight_mass] with make_drone's defaults for missing keys"""
    params = params or {}
    return [float(params.get(k, DEFAULT_PARAMS[k])) for k in PARAM_NAMES]


def make_drone(*args, **kwargs):
    raise NotImplementedError("no MJCF is built: the model constants are derived on the GPU (csrc/qd_model.h); "
                              "inspect them with env.model_constants()")


make_sim = mjcf_to_mjmodel = make_drone
