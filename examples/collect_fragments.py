#!/usr/bin/env python3
"""The rollout side of train_PPO.py / train_RMA.py without leaving the GPU: env + the reference's actor, exploring, PPO sample
batches of `fragment` steps, the statistics the reference logs about each batch, and (under torchrun) the per-fragment all-gather
that hands the trajectories to a central learner.

    python examples/collect_fragments.py                      # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/collect_fragments.py

A trained checkpoint goes in with  weights = mujoco_drone_amd.evaluation.load_policy_state(ckpt)["weights"]; here the network
is random-init (there are no checkpoints in the reference repository)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(num_envs=4096, fragment=256, fragments=4, family="RMA_full", quiet=False):
    import torch
    from mujoco_drone_amd import parallel as par
    from mujoco_drone_amd.custom_logging import BatchStatistics, EpisodeStatistics
    from mujoco_drone_amd.environments.BaseDroneEnv import base_config
    from mujoco_drone_amd.environments.observation_wrappers import LocalFrameRPYParamsEnv
    from mujoco_drone_amd.environments.rewards import distance_energy_reward
    from mujoco_drone_amd.policy import DevicePolicy, random_weights

    rank, world, local = par.init_distributed()
    device = "cuda:%d" % local
    torch.cuda.set_device(local)
    cfg = dict(base_config, num_drones=num_envs, reward_fcn=distance_energy_reward, random_params=True, param_difficulty=1,   # train_RMA.py:66-75
               state_difficulty=0.2, max_steps=1024, regen_env_at_steps=0, auto_reset=True, seed=par.shard_seed(42, rank), device=device)
    env = LocalFrameRPYParamsEnv(cfg)
    policy = DevicePolicy(family, random_weights(family, 0), device=device)
    batch_stats, episode_stats = BatchStatistics(), EpisodeStatistics(num_envs, device)
    obs, prev = env.vector_reset_tensor().clone(), None
    log = []
    for f in range(fragments):
        t0 = time.perf_counter()
        batch = policy.rollout(env._dev, fragment, obs, prev, explore=True, seed=1234 + rank, counter0=f * fragment, want_logp=True, want_value=True)
        result = batch_stats.on_learn_on_batch(train_batch={"obs": batch["obs"], "actions": batch["actions"]}, result={})
        info = episode_stats.update(batch["reward"], batch["truncated"])
        if world > 1:                                           # hand the fragment to rank 0's learner: one collective per tensor here
            gathered = [torch.empty((world,) + tuple(batch[k].shape), dtype=batch[k].dtype, device=device) for k in ("obs", "actions", "reward")]
            for g, k in zip(gathered, ("obs", "actions", "reward")):
                torch.distributed.all_gather_into_tensor(g.view((-1,) + tuple(batch[k].shape[1:])), batch[k])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        obs, prev = batch["obs"][-1].clone(), batch["actions"][-1].clone()
        log.append((info["episodes"], info["episode_reward_mean"], result["mean_obs0"], num_envs * fragment / dt))
        if rank == 0 and not quiet:
            print("fragment %d: %d episodes ended, mean return %.2f, mean length %.1f, mean |action| %.3f, %.2e env-steps/s"
                  % (f, info["episodes"], info["episode_reward_mean"], info["episode_len_mean"], result["mean_act0"], log[-1][3]))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return log


if __name__ == "__main__":
    main()
