"""rollout.py's dataset collection (rollout.py:64-86) on the GPU: batches of closed-loop policy rollouts
{'z': parameter embedding, 'o': observations, 'a': actions, 't': truncated flags}, written with pickle.

Per batch the reference regenerates the drone parameters (`reset_model(regen=True)`), rolls the policy out for
`rollout_length` steps with `policy.compute_actions(obs, prev_action_batch=prev_actions)` and reads `policy.model.z`
(the RMA networks' embedding of the env parameters) after the last step.  Here one batch is one `qd_rollout_policy`
call (fused kernel up to 4096 envs); `prev_actions` is carried from batch to batch as the reference does.
"""
import pickle

import numpy as np
import torch


def collect_dataset(env, policy, num_batches, rollout_length, explore=True, seed=0, as_lists=False):
    """env: a mirrored env object (e.g. LocalFrameRPYParamsEnv) with auto_reset; policy: DevicePolicy with an embedding.
    Returns a list of dicts with keys 'z' [N, 8], 'o' [T, N, D], 'a' [T, N, 4], 't' [T, N] (numpy); as_lists=True gives
    the reference's nesting (lists over time of arrays over envs)."""
    batches, prev, counter = [], None, 0
    for _ in range(int(num_batches)):
        env.reset_model(regen=True)                                   # rollout.py:69: new parameters, new initial states
        obs = env._dev.obs.clone()
        out = policy.rollout(env._dev, rollout_length, obs, prev_actions0=prev, explore=explore, seed=seed, counter0=counter)
        counter += int(rollout_length)
        prev = out["actions"][-1].clone()                             # :73 prev_actions persists across batches
        z = policy.embedding(out["obs"][-2] if rollout_length > 1 else obs, out["actions"][-2] if rollout_length > 1 else None)
        torch.cuda.synchronize()
        b = {"z": z.cpu().numpy(), "o": out["obs"].cpu().numpy(), "a": out["actions"].cpu().numpy(),
             "t": out["truncated"].cpu().numpy().astype(bool)}
        if as_lists:
            b = {"z": b["z"], "o": list(b["o"]), "a": list(b["a"]), "t": list(b["t"])}
        batches.append(b)
    return batches


def write_dataset(path, batches):
    """rollout.py:85-86"""
    with open(path, "wb") as f:
        pickle.dump(batches, f)
