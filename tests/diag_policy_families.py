"""diagnostic (not a test): forward time of every policy family at 4096 envs, specialised kernel vs interpreter"""
import os, sys, subprocess
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_drone_amd.policy import DevicePolicy, random_weights
FAMS = [("RMA_full", 22), ("RMA_model", 22), ("SimpleMLPmodel", 22), ("CustomMLP", 22), ("RMA_full_adapt", 22), ("CNNestimator", 23),
        ("CNNestimator_estimate", 23), ("LSTMestimator_estimate", 19), ("RMA_model_smaller", 22), ("RMA_model_smaller2", 22),
        ("CustomLSTM", 22), ("CustomLSTMbigger", 22), ("CustomLSTMbiggerCommonF", 22), ("DSN_LSTM_model", 22)]
n = 4096
for fam, D in FAMS:
    kw = dict(obs_dim=D, num_states=D if D != 22 else 16)
    w = random_weights(fam, 1)
    pol = DevicePolicy(fam, w, **kw)
    obs = torch.randn((n, D), device="cuda"); prev = torch.rand((n, 4), device="cuda"); out = torch.empty((n, 4), device="cuda")
    pol.reset_state(n)
    for k in range(30):
        pol.forward(obs, prev, out=out, counter=k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(200):
        pol.forward(obs, prev, out=out, counter=30 + k)
    e1.record(); torch.cuda.synchronize()
    print("%-24s kernel %d  forward %.2f us" % (fam, pol.kernel, e0.elapsed_time(e1) * 1000 / 200), flush=True)
