"""MI355X-native vectorised quadrotor(+hanging load) RL environment.

Drop-in for the env step path of TichyTech/mujoco-drone (environments/BaseDroneEnv.py,
observation_wrappers.py, rewards.py, SimpleDrone.py, transformation.py): the same
VectorEnv / gym surface, with the per-step work done by hand-written HIP kernels for
gfx950 behind the C ABI of include/qd.h.  There is no CPU compute path: importing the
environments without the built library raises.
"""
__all__ = ["build", "environments", "parallel"]  # build recipe: python -m mujoco_drone_amd.build
__version__ = "0.1.0"
