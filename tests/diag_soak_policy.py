"""diagnostic (not a test): soak of the closed-loop policy rollouts -- BASELINE config 3, 4096 envs, random-init networks,
exploring (sampled Beta actions, log-probabilities, value head) -- fused kernel (RMA_full) and the two-launch path with the
windowed adaptation network; invariants after every fragment: everything finite, actions in [0, 1], envs within bounds,
episode bookkeeping consistent with the truncation flags.   usage: python tests/diag_soak_policy.py [fragments]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mujoco_drone_amd.policy import DevicePolicy, random_weights
from mujoco_drone_amd.custom_logging import BatchStatistics, EpisodeStatistics

frags = int(sys.argv[1]) if len(sys.argv) > 1 else 100
T = 1024
for fam, conf, n, kw in (("RMA_full", "config3", 4096, {}), ("RMA_full_adapt", "config3", 4096, {}),
                         ("CNNestimator", "config5", 8192, dict(obs_dim=23, num_states=23))):   # (the last: 32 envs per workgroup)
    env, _ = bench.make_env(conf, n, 42, "cuda:0")
    pol = DevicePolicy(fam, random_weights(fam, 3), **kw)
    obs0 = env.vector_reset_tensor().clone()
    if pol.has_history:
        pol.reset_state(n)
    es, bs = EpisodeStatistics(n), BatchStatistics()
    prev = None
    counter = 0
    t0 = time.perf_counter()
    episodes = 0
    for f in range(frags):
        out = pol.rollout(env._dev, T, obs0, prev, explore=True, seed=7, counter0=counter, want_logp=True, want_value=True)
        obs, act, rew, tr, logp, val = out["obs"], out["actions"], out["reward"], out["truncated"], out["logp"], out["value"]
        counter += T
        info = es.update(rew, tr)
        st = bs.column_stats_tensor(act)
        ok = all(bool(torch.isfinite(x).all()) for x in (obs, act, rew, logp, val))
        amin, amax = float(st[0].min()), float(st[1].max())
        assert ok and 0.0 <= amin and amax <= 1.0, (fam, f, ok, amin, amax)
        assert info["episodes"] == int(tr.sum()), (info["episodes"], int(tr.sum()))
        episodes += info["episodes"]
        obs0, prev = obs[-1].clone(), act[-1].clone()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    q = env._dev.get_state()[0]
    dist = float((q[:, :3] - torch.tensor([0, 0, 15.0], device=q.device)).norm(dim=1).max()) if conf == "config3" else 0.0   # (config 5's waypoint circles)
    print("%-16s kernel %d: %d fragments x %d steps x %d envs = %.2e env-steps in %.1f s (%.2e /s incl. checks), %d episodes, max |pos - ref| %.3f"
          % (fam, pol.kernel, frags, T, n, frags * T * n, dt, frags * T * n / dt, episodes, dist), flush=True)
    assert dist < 4.5
print("policy soak ok")
