"""Diagnostic (not a test): a run-time-dispatched configuration (LocalFrameRmParamsEnv + reward_3) through the persistent fragment
kernel at batches of two workgroups per CU: us per step.   usage: python tests/diag_generic_big.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_drone_amd import parallel as par
from mujoco_drone_amd.environments import observation_wrappers as ow, rewards
from mujoco_drone_amd.environments.BaseDroneEnv import base_config
for n, T in ((65536, 128), (1048576, 32)):
    cfg = dict(base_config)
    cfg.update(num_drones=n, random_params=True, param_difficulty=1, state_difficulty=0.2, max_steps=1024, regen_env_at_steps=10 ** 9,
               reward_fcn=rewards.reward_3, seed=42, device="cuda:0", auto_reset=True)
    env = ow.LocalFrameRmParamsEnv(cfg)
    env.vector_reset_tensor()
    f = par.FragmentBuffers(T, n, env._dev.D, "cuda:0")
    f.actions.copy_(torch.rand(f.actions.shape, device="cuda"))
    for _ in range(3):
        env.step_fragment_tensor(f.actions, f.obs, f.rewards, f.truncated)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        env.step_fragment_tensor(f.actions, f.obs, f.rewards, f.truncated)
    e1.record(); torch.cuda.synchronize()
    print("%s n=%d: %.2f us per step" % (env._dev.fragment_kernel_name(), n, e0.elapsed_time(e1) * 1e3 / (6 * T)), flush=True)
    del env, f; torch.cuda.empty_cache()
