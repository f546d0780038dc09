#!/usr/bin/env python3
"""Generate tests/golden/policy_vectors.npz from the reference's own policy classes.

Runs ONLY in the build container (needs /root/reference and torch); never on the GPU box.
The reference's models (models/PPO/RMA/RMA_model.py: RMA_full, RMA_model; models/PPO/SimpleMLP/SimpleMLP.py:
SimpleMLPmodel) and its action distribution (distributions.py: MyBetaDist) are imported from where they lie and
run in eval mode on seeded inputs; weights, inputs and outputs are stored as data.

ray is absent here.  Unlike the placeholders of make_golden.py, four of the ray names these files use carry
behaviour the forward pass depends on, so they are replaced by FUNCTIONAL stand-ins written from ray's documented
behaviour: SlimFC = nn.Linear followed by the named activation (state-dict keys `<x>._model.0.weight/bias`, the
layout of the reference's checkpoints), normc_initializer, TorchModelV2.__init__ (stores its arguments),
TorchDistributionWrapper.__init__ (stores inputs / model).  These vectors therefore pin the reference's WIRING
(slicing, concatenation order, layer sizes, BatchNorm in eval mode, the Beta parameterisation), not ray's layers.
Nothing from /root/reference is copied: the output holds numbers only.
"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "policy_vectors.npz")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import _install_placeholders  # noqa: E402


def _install_ray_standins():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class TorchModelV2:
        def __init__(self, obs_space, action_space, num_outputs, model_config, name):
            self.obs_space, self.action_space, self.num_outputs = obs_space, action_space, num_outputs
            self.model_config, self.name = model_config, name
            self.view_requirements = {}

    def normc_initializer(std=1.0):
        def init(tensor):
            tensor.data.normal_(0, 1)
            tensor.data *= std / torch.sqrt(tensor.data.pow(2).sum(1, keepdim=True))
        return init

    class SlimFC(nn.Module):
        def __init__(self, in_size, out_size, initializer=None, activation_fn=None, use_bias=True, bias_init=0.0):
            super().__init__()
            linear = nn.Linear(in_size, out_size, bias=use_bias)
            (initializer or nn.init.xavier_uniform_)(linear.weight)
            if use_bias:
                nn.init.constant_(linear.bias, bias_init)
            layers = [linear]
            if activation_fn == 'tanh':
                layers.append(nn.Tanh())
            elif activation_fn == 'relu':
                layers.append(nn.ReLU())
            elif activation_fn is not None and activation_fn != 'linear':
                raise ValueError(activation_fn)
            self._model = nn.Sequential(*layers)

        def forward(self, x):
            return self._model(x)

    class AppendBiasLayer(nn.Module):
        pass

    class ViewRequirement:
        def __init__(self, *a, **k):
            self.args, self.kw = a, k

    class SampleBatch(dict):
        is_training = False

    class TorchDistributionWrapper:
        def __init__(self, inputs, model):
            self.inputs, self.model = inputs, model

    class TorchBeta(TorchDistributionWrapper):
        pass

    import typing
    mod("ray.rllib.models"); mod("ray.rllib.models.torch")
    mod("ray.rllib.models.torch.torch_modelv2", TorchModelV2=TorchModelV2)
    mod("ray.rllib.models.torch.misc", SlimFC=SlimFC, AppendBiasLayer=AppendBiasLayer, normc_initializer=normc_initializer)
    mod("ray.rllib.models.torch.torch_action_dist", TorchBeta=TorchBeta, TorchDistributionWrapper=TorchDistributionWrapper)
    mod("ray.rllib.utils"); mod("ray.rllib.utils.annotations", override=lambda cls: (lambda f: f))
    mod("ray.rllib.utils.framework", try_import_torch=lambda: (torch, nn))
    mod("ray.rllib.utils.typing", Dict=typing.Dict, TensorType=typing.Any, List=typing.List, ModelConfigDict=dict)
    mod("ray.rllib.policy"); mod("ray.rllib.policy.sample_batch", SampleBatch=SampleBatch)
    mod("ray.rllib.policy.view_requirement", ViewRequirement=ViewRequirement)
    mod("ray.rllib.models.torch.recurrent_net", RecurrentNetwork=type("RecurrentNetwork", (TorchModelV2,), {}))   # only subclassed
    mod("ray.rllib.policy.rnn_sequencing", add_time_dimension=None)                                              # unused on this path


def _randomise(model, gen):
    """trained-looking weights: non-zero biases, non-trivial BatchNorm statistics"""
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=gen) * 0.1)
        for m in model.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=gen) * 0.2)
                m.running_var.copy_(torch.rand(m.num_features, generator=gen) * 1.5 + 0.25)
                m.weight.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.num_features, generator=gen) * 0.1)


def main():
    _install_placeholders()
    _install_ray_standins()
    sys.path.insert(0, REF)
    import gymnasium
    from models.PPO.RMA.RMA_model import RMA_full, RMA_model
    from models.PPO.SimpleMLP.SimpleMLP import SimpleMLPmodel
    from models.PPO.MLP.CustomMLP import CustomMLP
    from distributions import MyBetaDist

    gen = torch.Generator().manual_seed(20250614)
    n, D = 48, 22
    obs_space = gymnasium.spaces.Box(low=-np.inf, high=np.inf, shape=(D,))
    act_space = gymnasium.spaces.Box(low=0, high=1, shape=(4,))
    cc = {'num_states': 16, 'num_params': 6, 'num_actions': 4, 'param_embed_dim': 8, 'train_adaptation': False,
          'adapt_seq_len': 32}                                   # train_PPO.py:39-45
    obs = torch.randn((n, D), generator=gen) * 1.5
    obs[:, 16:] = torch.tensor([1, 0.17, 7, 0.01, 1.2, 0.3]) * (1 + 0.1 * torch.randn((n, 6), generator=gen))
    prev = torch.rand((n, 4), generator=gen)
    out = {"obs": obs.numpy(), "prev_actions": prev.numpy()}
    for tag, cls in (("rma_full", RMA_full), ("rma_model", RMA_model), ("simple_mlp", SimpleMLPmodel), ("custom_mlp", CustomMLP)):
        torch.manual_seed(7)
        model = cls(obs_space, act_space, 8, {"custom_model_config": cc}, tag)
        _randomise(model, gen)
        model.eval()
        if cls is RMA_full:
            inp = {"obs_history": obs, "action_history": prev, "is_training": False}
        else:
            inp = {"obs": obs, "prev_actions": prev, "is_training": False}
        with torch.no_grad():
            logits, _ = model.forward(inp, [], None)
            value = model.value_function()
            dist = MyBetaDist(logits, model)
            action = dist.deterministic_sample()
            logp = dist.logp(action)
        out[tag + "_logits"], out[tag + "_value"] = logits.numpy(), value.numpy()
        out[tag + "_action"], out[tag + "_logp"] = action.numpy(), logp.numpy()
        # the adaptation module is not on this path (train_adaptation=False, train_PPO.py:43): its weights stay out
        sd = {k: v for k, v in model.state_dict().items() if not k.startswith("adaptation_module")}
        out[tag + "_keys"] = np.array(list(sd.keys()))
        for k, v in sd.items():
            out[tag + "/" + k] = v.numpy()
    # RMA_full with the adaptation module in the loop (train_RMA.py:39-45: train_adaptation=True, adapt_seq_len=32):
    # z_hat = TimeCNN2(32-step history of (state, previous action)) replaces the parameter encoding
    cca = dict(cc, train_adaptation=True)
    torch.manual_seed(11)
    model = RMA_full(obs_space, act_space, 8, {"custom_model_config": cca}, "rma_adapt")
    _randomise(model, gen)
    with torch.no_grad():                                         # Conv1d biases start at torch's default; make them visible
        for m in model.modules():
            if isinstance(m, nn.Conv1d):
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.1)
    model.eval()
    nh, L = 24, 32
    obs_h = torch.randn((nh, L, D), generator=gen) * 1.2
    act_h = torch.rand((nh, L, 4), generator=gen)
    obs_h[:8, :20] = 0.0                                          # episodes younger than the window: zero-padded rows
    act_h[:8, :21] = 0.0
    with torch.no_grad():
        logits, _ = model.forward({"obs_history": obs_h, "action_history": act_h, "is_training": False}, [], None)
        value = model.value_function()
        z_hat = model.z_hat
    out["rma_adapt_obs_history"], out["rma_adapt_action_history"] = obs_h.numpy(), act_h.numpy()
    out["rma_adapt_logits"], out["rma_adapt_value"], out["rma_adapt_z_hat"] = logits.numpy(), value.numpy(), z_hat.numpy()
    sd = model.state_dict()
    out["rma_adapt_keys"] = np.array(list(sd.keys()))
    for k, v in sd.items():
        out["rma_adapt/" + k] = v.numpy()
    # train_LSTM.py's network (BASELINE config 5): CNNestimator with use_estimate=False, train_estimator=False -- feed-forward
    # on the 23-value LocalFrameFullStateEnv observation; and with use_estimate=True: the pendulum state is replaced by the
    # TimeCNN estimate from the 32-step history
    from models.PPO.CustomLSTM.StateEstimatorLSTM import CNNestimator
    D5 = 23
    obs_space5 = gymnasium.spaces.Box(low=-np.inf, high=np.inf, shape=(D5,))
    obs5 = torch.randn((n, D5), generator=gen) * 1.3
    out["obs23"] = obs5.numpy()
    for tag, use_est in (("cnn_est_ff", False), ("cnn_est_hist", True)):
        torch.manual_seed(13)
        model = CNNestimator(obs_space5, act_space, 8, {"custom_model_config": {'num_states': 23, 'num_actions': 4, 'use_estimate': use_est,
                                                                               'train_estimator': False}, "max_seq_len": 32}, tag)
        _randomise(model, gen)
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, nn.Conv1d):
                    m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.1)
        model.eval()
        if use_est:
            oh = torch.randn((nh, L, D5), generator=gen) * 1.2
            ah = torch.rand((nh, L, 4), generator=gen)
            oh[:8, :20] = 0.0
            ah[:8, :21] = 0.0
            inp = {"obs_history": oh, "action_history": ah, "is_training": False}
            out[tag + "_obs_history"], out[tag + "_action_history"] = oh.numpy(), ah.numpy()
        else:
            inp = {"obs_history": obs5, "action_history": prev, "is_training": False}
        with torch.no_grad():
            logits, _ = model.forward(inp, [], None)
            value = model.value_function()
        out[tag + "_logits"], out[tag + "_value"] = logits.numpy(), value.numpy()
        if use_est:
            out[tag + "_estimate"] = model.pendulum_state_estimate.numpy()
        sd = {k: v for k, v in model.state_dict().items() if use_est or not k.startswith("estimation_module")}
        out[tag + "_keys"] = np.array(list(sd.keys()))
        for k, v in sd.items():
            out[tag + "/" + k] = v.numpy()
    # LSTMestimator (StateEstimatorLSTM.py:15-147) with the nn.LSTM pendulum-state estimator in the loop.  Its forward() only
    # assembles `inputs` from the view requirements (prev_o = [o_{t-1}, o_t], prev_a) through ray's add_time_dimension; the
    # network proper is forward_rnn(inputs [B, T, 34], state), called here directly on explicit sequences.
    from models.PPO.CustomLSTM.StateEstimatorLSTM import LSTMestimator
    D19 = 19
    obs_space19 = gymnasium.spaces.Box(low=-np.inf, high=np.inf, shape=(D19,))
    torch.manual_seed(17)
    model = LSTMestimator(obs_space19, act_space, 8, {"custom_model_config": {'num_states': 19, 'num_actions': 4, 'use_estimate': True,
                                                                              'train_estimator': False}}, "lstm_est")
    _randomise(model, gen)
    with torch.no_grad():
        for name, prm in model.estimation_module.LSTM.named_parameters():
            if "bias" in name:
                prm.copy_(torch.randn(prm.shape, generator=gen) * 0.1)
    model.eval()
    Bn, Tn = 12, 24
    o_seq = torch.randn((Bn, Tn, D19), generator=gen) * 1.2
    a_seq = torch.rand((Bn, Tn, 4), generator=gen)                     # a_seq[:, t] = the action taken after o_seq[:, t]
    o_prev = torch.cat([torch.zeros((Bn, 1, 15)), o_seq[:, :-1, :15]], dim=1)       # zero before the episode start
    a_prev = torch.cat([torch.zeros((Bn, 1, 4)), a_seq[:, :-1]], dim=1)
    inputs = torch.cat([o_prev, o_seq[:, :, :15], a_prev], dim=-1)                   # :77-78
    model.gt_pendulum_states = o_seq[:, :, 15:]
    with torch.no_grad():
        logits, state_out = model.forward_rnn(inputs, [torch.zeros((Bn, 32)), torch.zeros((Bn, 32))], None, False)
        value = model.value_function()
    out["lstm_est_obs_seq"], out["lstm_est_action_seq"] = o_seq.numpy(), a_seq.numpy()
    out["lstm_est_logits"], out["lstm_est_value"] = logits.numpy(), value.numpy().reshape(Bn, Tn)
    out["lstm_est_estimates"] = model.pendulum_state_estimates.numpy()
    out["lstm_est_h"], out["lstm_est_c"] = state_out[0].numpy(), state_out[1].numpy()
    sd = model.state_dict()
    out["lstm_est_keys"] = np.array(list(sd.keys()))
    for k, v in sd.items():
        out["lstm_est/" + k] = v.numpy()
    # CustomLSTM (models/PPO/CustomLSTM/CustomLSTM.py:14-105): the LSTM sits in the action path; forward_rnn on explicit sequences
    from models.PPO.CustomLSTM.CustomLSTM import CustomLSTM
    torch.manual_seed(19)
    model = CustomLSTM(obs_space, act_space, 8, {"custom_model_config": {'num_states': 22, 'num_params': 0, 'num_actions': 4}}, "custom_lstm")
    _randomise(model, gen)
    with torch.no_grad():
        for name, prm in model.LSTM.named_parameters():
            if "bias" in name:
                prm.copy_(torch.randn(prm.shape, generator=gen) * 0.1)
    model.eval()
    o_seq = torch.randn((Bn, Tn, D), generator=gen) * 1.2
    a_seq = torch.rand((Bn, Tn, 4), generator=gen)
    a_prev = torch.cat([torch.zeros((Bn, 1, 4)), a_seq[:, :-1]], dim=1)
    with torch.no_grad():
        logits, state_out = model.forward_rnn(torch.cat([o_seq, a_prev], dim=-1), [torch.zeros((Bn, 64)), torch.zeros((Bn, 64))], None, False)
        value = model.value_function()
    out["custom_lstm_obs_seq"], out["custom_lstm_action_seq"] = o_seq.numpy(), a_seq.numpy()
    out["custom_lstm_logits"], out["custom_lstm_value"] = logits.numpy(), value.numpy().reshape(Bn, Tn)
    sd = model.state_dict()
    out["custom_lstm_keys"] = np.array(list(sd.keys()))
    for k, v in sd.items():
        out["custom_lstm/" + k] = v.numpy()
    # the remaining variants the training scripts import (train_PPO.py:7-10, evaluation.py:7): RMA_model_smaller and
    # RMA_model_smaller2 (the second definition, RMA_model.py:398-437, is the one Python keeps; residual blocks in the value
    # head), same inputs as rma_model
    from models.PPO.RMA.RMA_model import RMA_model_smaller, RMA_model_smaller2
    for tag, cls, seed in (("rma_smaller", RMA_model_smaller, 23), ("rma_smaller2", RMA_model_smaller2, 29)):
        torch.manual_seed(seed)
        model = cls(obs_space, act_space, 8, {"custom_model_config": cc}, tag)
        _randomise(model, gen)
        model.eval()
        with torch.no_grad():
            logits, _ = model.forward({"obs": obs, "prev_actions": prev, "is_training": False}, [], None)
            value = model.value_function()
        out[tag + "_logits"], out[tag + "_value"], out[tag + "_z"] = logits.numpy(), value.numpy(), model.z.numpy()
        sd = model.state_dict()
        out[tag + "_keys"] = np.array(list(sd.keys()))
        for k, v in sd.items():
            out[tag + "/" + k] = v.numpy()
    # CustomLSTMbigger / CustomLSTMbiggerCommonF (CustomLSTM.py:107-299) and DSN_LSTM_model (DSN_LSTM_model.py:20-160):
    # forward_rnn on explicit 24-step episodes from zero state
    from models.PPO.CustomLSTM.CustomLSTM import CustomLSTMbigger, CustomLSTMbiggerCommonF
    from models.PPO.DSN_LSTM.DSN_LSTM_model import DSN_LSTM_model
    for tag, cls, seed in (("lstm_bigger", CustomLSTMbigger, 31), ("lstm_common_f", CustomLSTMbiggerCommonF, 37)):
        torch.manual_seed(seed)
        model = cls(obs_space, act_space, 8, {"custom_model_config": {'num_states': 22, 'num_params': 0, 'num_actions': 4}}, tag)
        _randomise(model, gen)
        with torch.no_grad():
            for name, prm in model.LSTM.named_parameters():
                if "bias" in name:
                    prm.copy_(torch.randn(prm.shape, generator=gen) * 0.1)
        model.eval()
        o_seq = torch.randn((Bn, Tn, D), generator=gen) * 1.2
        a_seq = torch.rand((Bn, Tn, 4), generator=gen)
        a_prev = torch.cat([torch.zeros((Bn, 1, 4)), a_seq[:, :-1]], dim=1)
        with torch.no_grad():
            logits, state_out = model.forward_rnn(torch.cat([o_seq, a_prev], dim=-1), [torch.zeros((Bn, 64)), torch.zeros((Bn, 64))], None, False)
            value = model.value_function()
        out[tag + "_obs_seq"], out[tag + "_action_seq"] = o_seq.numpy(), a_seq.numpy()
        out[tag + "_logits"], out[tag + "_value"] = logits.numpy(), value.numpy().reshape(Bn, Tn)
        sd = model.state_dict()
        out[tag + "_keys"] = np.array(list(sd.keys()))
        for k, v in sd.items():
            out[tag + "/" + k] = v.numpy()
    torch.manual_seed(41)
    model = DSN_LSTM_model(obs_space, act_space, 8, {"custom_model_config": cc}, "dsn_lstm")
    _randomise(model, gen)
    with torch.no_grad():
        for lstm in (model.LSTM_x, model.LSTM_y, model.LSTM_z):
            for name, prm in lstm.named_parameters():
                if "bias" in name:
                    prm.copy_(torch.randn(prm.shape, generator=gen) * 0.1)
    model.eval()
    o_seq = torch.randn((Bn, Tn, D), generator=gen) * 1.2
    a_seq = torch.rand((Bn, Tn, 4), generator=gen)
    a_prev = torch.cat([torch.zeros((Bn, 1, 4)), a_seq[:, :-1]], dim=1)
    with torch.no_grad():
        logits, state_out = model.forward_rnn(o_seq, a_prev, [torch.zeros((Bn, k)) for k in (32, 32, 32, 32, 16, 16)], None, False)
        value = model.value_function()
    out["dsn_lstm_obs_seq"], out["dsn_lstm_action_seq"] = o_seq.numpy(), a_seq.numpy()
    out["dsn_lstm_logits"], out["dsn_lstm_value"] = logits.numpy(), value.numpy().reshape(Bn, Tn)
    sd = model.state_dict()
    out["dsn_lstm_keys"] = np.array(list(sd.keys()))
    for k, v in sd.items():
        out["dsn_lstm/" + k] = v.numpy()
    # MySquashedGaussian (distributions.py:41-119) on seeded logits: the deterministic action, and logp of given actions
    from distributions import MySquashedGaussian
    sg_logits = torch.randn((64, 8), generator=gen) * torch.tensor([1.5] * 4 + [2.5] * 4)
    sg_logits[0, 4:] = torch.tensor([-9.0, 9.0, -5.0, 5.0])        # log_std beyond / at the clamp
    sg_x = torch.rand((64, 4), generator=gen)
    sg_x[1] = torch.tensor([0.0, 1.0, 1e-6, 1 - 1e-6])             # actions at the edge of the squashing range
    with torch.no_grad():
        dist = MySquashedGaussian(sg_logits, None)
        sg_action = dist.deterministic_sample()
        out["sg_logits"], out["sg_x"] = sg_logits.numpy(), sg_x.numpy()
        out["sg_action"], out["sg_logp_action"], out["sg_logp_x"] = sg_action.numpy(), dist.logp(sg_action).numpy(), dist.logp(sg_x).numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "keys:", len(out), "bytes:", os.path.getsize(OUT))


if __name__ == "__main__":
    main()
