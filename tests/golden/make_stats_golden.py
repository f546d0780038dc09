#!/usr/bin/env python3
"""Generate tests/golden/stats_vectors.npz from the reference's own train-batch callback.

Runs ONLY in the build container (needs /root/reference and torch).  custom_logging.py is imported from where it lies, with
empty placeholder classes for the two ray names it subclasses / mentions (DefaultCallbacks, UnifiedLogger: neither carries
behaviour on this path); MyCallbacks.on_learn_on_batch (custom_logging.py:9-31) is then run on a seeded batch and the numbers it
writes into `result` are stored next to the inputs.  Nothing from /root/reference is copied: the output holds numbers only."""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stats_vectors.npz")


def main():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    for name in ("ray", "ray.rllib", "ray.rllib.algorithms", "ray.tune"):
        mod(name)
    mod("ray.rllib.algorithms.callbacks", DefaultCallbacks=type("DefaultCallbacks", (), {}))
    mod("ray.tune.logger", UnifiedLogger=type("UnifiedLogger", (), {}))
    sys.path.insert(0, REF)
    from custom_logging import MyCallbacks

    gen = torch.Generator().manual_seed(20250614)
    out = {}
    for tag, rows in (("small", 37), ("batch", 3000)):
        obs = torch.randn((rows, 22), generator=gen) * torch.linspace(0.1, 3.0, 22) + torch.linspace(-2, 15, 22)
        actions = torch.rand((rows, 4), generator=gen)
        result = {}
        MyCallbacks().on_learn_on_batch(policy=None, train_batch={"obs": obs, "actions": actions}, result=result)
        out[tag + "_obs"], out[tag + "_actions"] = obs.numpy(), actions.numpy()
        for what, cols in (("obs", 22), ("act", 4)):
            for stat in ("min", "max", "mean", "var"):
                out["%s_%s_%s" % (tag, stat, what)] = np.array([result["%s_%s%d" % (stat, what, i)] for i in range(cols)], dtype=np.float64)
        assert len(result) == 4 * (22 + 4)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "keys:", len(out), "bytes:", os.path.getsize(OUT))


if __name__ == "__main__":
    main()
