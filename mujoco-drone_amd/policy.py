"""On-device policy inference: the reference's actor / critic networks as layer programs for
`qd_policy_*` (include/qd.h), compiled from a state dict keyed like the reference's checkpoints
(`policies/default_policy/policy_state.pkl` -> 'weights', evaluation.py:155-159).

Families (eval mode; constructor arguments as in train_PPO.py:39-45):
  "RMA_full"        models/PPO/RMA/RMA_model.py:17-110 with train_adaptation=False
  "RMA_model"       models/PPO/RMA/RMA_model.py:199-292
  "SimpleMLPmodel"  models/PPO/SimpleMLP/SimpleMLP.py:18-98
  "CustomMLP"       models/PPO/MLP/CustomMLP.py:17-98
  "CNNestimator"    models/PPO/CustomLSTM/StateEstimatorLSTM.py:200-283 with use_estimate=False (train_LSTM.py:51-60; obs_dim =
                    num_states = 23); "CNNestimator_estimate": use_estimate=True, TimeCNN over the 32-step history, incremental
  "CustomLSTM"      models/PPO/CustomLSTM/CustomLSTM.py:14-105 (recurrent actor: nn.LSTM(64, 64) in the action path)
  "LSTMestimator"   models/PPO/CustomLSTM/StateEstimatorLSTM.py:15-147 (obs_dim 19), use_estimate=False; "LSTMestimator_estimate":
                    the nn.LSTM pendulum-state estimator in the loop (h, c and the previous observation kept per env)
  "RMA_full_adapt"  RMA_full with train_adaptation=True, adapt_seq_len=32 (train_RMA.py:39-45): the adaptation CNN over the
                    32-step history, evaluated incrementally from per-env rings (pass consecutive `counter`s; call
                    reset_state at the start; envs flagged in prev_truncated restart their history by themselves)
The deterministic action is MyBetaDist's (distributions.py:8-26).  There is no CPU path: everything runs in
libqd.so's k_policy kernel.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L

BN_EPS = 1e-5  # torch.nn.BatchNorm1d default, what the reference's models use


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _Program:
    """accumulates ops and the flat float32 weight blob they index"""

    def __init__(self, weights):
        self.w = {k: np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v) for k, v in weights.items()}
        self.ops, self.blob, self.n = [], [], 0
        self.flags = 0   # set to POL_VALUE_ONLY while the value head's ops are emitted
        self.rings = []  # (rows, width, period, fill offset)

    def _put(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32).ravel()
        off = self.n
        self.blob.append(arr)
        self.n += arr.size
        return off

    def copy_obs(self, src_off, count, buf, off):
        self.ops.append(L.QdPolicyOp(L.POL_COPY_OBS, 0, src_off, count, buf, off, count, 0, self.flags, 0, 0, 0))

    def copy_prev(self, count, buf, off):
        self.ops.append(L.QdPolicyOp(L.POL_COPY_PREV, 0, 0, count, buf, off, count, 0, self.flags, 0, 0, 0))

    def fc(self, prefix, src, dst, act):
        """SlimFC `prefix`: (buf, off) -> (buf, off)"""
        W, b = self.w[prefix + "._model.0.weight"], self.w[prefix + "._model.0.bias"]
        out_dim, in_dim = W.shape
        self.ops.append(L.QdPolicyOp(L.POL_DENSE, src[0], src[1], in_dim, dst[0], dst[1], out_dim,
                                     {None: L.ACT_NONE, "tanh": L.ACT_TANH, "relu": L.ACT_RELU}[act], self.flags, 0,
                                     self._put(W), self._put(b)))
        return out_dim

    def dense(self, W, b, src, dst, act):
        """explicit weight matrix W [out, in] / bias b"""
        out_dim, in_dim = W.shape
        self.ops.append(L.QdPolicyOp(L.POL_DENSE, src[0], src[1], in_dim, dst[0], dst[1], out_dim,
                                     {None: L.ACT_NONE, "tanh": L.ACT_TANH, "relu": L.ACT_RELU}[act], self.flags, 0,
                                     self._put(W), self._put(b)))
        return out_dim

    def ring(self, rows, width, period, fill):
        """declare a history ring; returns its index"""
        self.rings.append((int(rows), int(width), int(period), self._put(np.asarray(fill, dtype=np.float32))))
        return len(self.rings) - 1

    def ring_load(self, ring, dst):
        rows, width = self.rings[ring][0], self.rings[ring][1]
        self.ops.append(L.QdPolicyOp(L.POL_RING_LOAD, ring, 0, rows * width, dst[0], dst[1], rows * width, 0, self.flags, 0, 0, 0))

    def ring_push(self, src, ring):
        width = self.rings[ring][1]
        self.ops.append(L.QdPolicyOp(L.POL_RING_PUSH, src[0], src[1], width, ring, 0, width, 0, self.flags, 0, 0, 0))

    def lstm_cell(self, gates, hc, hidden):
        """gates (buf, off) [4H] -> h' at hc (buf, off) [H], c updated in place right behind it"""
        self.ops.append(L.QdPolicyOp(L.POL_LSTM_CELL, gates[0], gates[1], 4 * hidden, hc[0], hc[1], hidden, 0, self.flags, 0, 0, 0))

    def bn(self, prefix, buf, off):
        """eval-mode BatchNorm1d `prefix` (or the concatenation of several, given as a tuple of prefixes) in place"""
        cat = lambda key: np.concatenate([self.w[q + key].astype(np.float64) for q in ((prefix,) if isinstance(prefix, str) else prefix)])
        g, b, m, v = cat(".weight"), cat(".bias"), cat(".running_mean"), cat(".running_var")
        scale = g / np.sqrt(v + BN_EPS)
        self.ops.append(L.QdPolicyOp(L.POL_AFFINE, buf, off, len(g), buf, off, len(g), 0, self.flags, 0, self._put(scale),
                                     self._put(b - m * scale)))


def _rma_full(p, D, ns, npar, na):
    X, P, A, B = 0, 1, 2, 3
    p.copy_obs(0, ns, X, 0); p.copy_prev(na, X, ns); p.copy_obs(D - npar, npar, P, 0)      # RMA_model.py:93-96
    h = p.fc("param_encoder.0", (P, 0), (A, 0), "tanh")
    z = p.fc("param_encoder.1", (A, 0), (X, ns + na), None)                                 # :107 z = param_encoder(e)
    p.fc("_hidden_layers.0", (X, 0), (A, 0), "tanh")                                        # :108 cat(flat_in, z)
    f = p.fc("_hidden_layers.1", (A, 0), (B, 0), "tanh")
    p.bn("_hidden_layers.2", B, 0)
    m = p.fc("_logits.0", (B, 0), (A, 0), "tanh")
    nl = p.fc("_logits.1", (A, 0), (P, 0), None)
    p.flags = L.POL_VALUE_ONLY
    m = p.fc("_value_branch.0", (B, 0), (A, 0), "tanh")
    p.fc("_value_branch.1", (A, 0), (A, 128 if m <= 128 else 256), "tanh")
    p.fc("_value_branch.2", (A, 128 if m <= 128 else 256), (X, 0), None)
    return dict(widths=[max(32, ns + na + z), 16, 512 if m > 128 else 256, max(f, 128)], logits=(P, 0, nl), value=(X, 0),
                aux=(X, ns + na, z))   # self.z, the parameter embedding (RMA_model.py:107)


def _rma_model(p, D, ns, npar, na):
    X, P, A, B = 0, 1, 2, 3
    p.copy_obs(0, ns, X, 0); p.copy_prev(na, X, ns); p.copy_obs(ns, npar, P, 0)            # RMA_model.py:273-285
    p.fc("param_encoder.0", (P, 0), (A, 0), "tanh")
    z = p.fc("param_encoder.1", (A, 0), (X, ns + na), "tanh")
    p.fc("_hidden_layers.0", (X, 0), (A, 0), "tanh")
    p.fc("_hidden_layers.1", (A, 0), (B, 0), "tanh")
    p.fc("_hidden_layers.2", (B, 0), (A, 0), "tanh")
    p.fc("_hidden_layers.3", (A, 0), (B, 0), "tanh")
    p.bn("_hidden_layers.4", B, 0)
    p.fc("_logits.0", (B, 0), (A, 0), "tanh")
    p.fc("_logits.1", (A, 0), (A, 64), "tanh")
    nl = p.fc("_logits.2", (A, 64), (P, 0), None)
    p.flags = L.POL_VALUE_ONLY
    p.fc("_value_branch.0", (B, 0), (A, 0), "tanh")
    p.fc("_value_branch.1", (A, 0), (A, 128), "tanh")
    p.fc("_value_branch.2", (A, 128), (X, 0), None)
    return dict(widths=[max(32, ns + na + z), 16, 256, 128], logits=(P, 0, nl), value=(X, 0), aux=(X, ns + na, z))   # self.z (:286)


def _simple_mlp(p, D, ns, npar, na):
    X, P, A, B = 0, 1, 2, 3
    nl = 0
    for trunk, buf in (("_logits", X), ("_value_branch", P)):                              # SimpleMLP.py:86-98
        p.flags = 0 if trunk == "_logits" else L.POL_VALUE_ONLY
        p.copy_obs(0, D, buf, 0); p.copy_prev(na, buf, D)
        p.bn(trunk + ".0", buf, 0)
        p.fc(trunk + ".1", (buf, 0), (A, 0), "tanh")
        p.fc(trunk + ".2", (A, 0), (B, 0), "tanh")
        p.fc(trunk + ".3", (B, 0), (A, 0), "tanh")
        p.fc(trunk + ".4", (A, 0), (B, 0), "tanh")
        p.bn(trunk + ".5", B, 0)
        p.fc(trunk + ".6", (B, 0), (A, 0), "tanh")
        p.fc(trunk + ".7", (A, 0), (A, 128), "tanh")
        n = p.fc(trunk + ".8", (A, 128), (buf, 0), None)
        nl = n if trunk == "_logits" else nl
    return dict(widths=[32, 32, 256, 128], logits=(X, 0, nl), value=(P, 0))


def _f32_fc(W, b, x, tanh):
    y = (np.asarray(W, np.float32) @ np.asarray(x, np.float32) + np.asarray(b, np.float32)).astype(np.float32)
    return np.tanh(y).astype(np.float32) if tanh else y


def _rma_full_adapt(p, D, ns, npar, na):
    """RMA_full with train_adaptation=True (RMA_model.py:77-110, TimeCNN2 :155-191; train_RMA.py:39-45, adapt_seq_len 32).

    The reference re-runs the whole adaptation CNN on the 32-step window every step.  Its pieces are functions of fixed
    absolute time windows -- inMLP acts per time step, Conv1d(32,32,5,stride 2) output p covers rows 2p..2p+4 of the
    window, Conv1d(32,16,5) output q covers conv1 outputs q..q+4 -- and the window slides by one row per step, so a
    conv1 / conv2 output computed at step t is the neighbouring output of step t+2.  The program therefore keeps three
    per-env rings (the last 5 inMLP rows; per step parity the last 4 conv1 and the last 9 conv2 outputs), computes ONE new
    value of each per step (20->32->32->32, 160->32, 160->16) and reads the rest back: same numbers, 14x / 10x fewer
    convolution MACs, 2.8 KB of history per env instead of a [32, 20] observation window plus recomputation.  Rows
    before the episode start are zero in the reference (RLlib's zero padding); their images (inMLP(0) and what the
    convolutions make of it) are the rings' episode-start fill values."""
    X, P, A, B, H, C1, C2 = 0, 1, 2, 3, 4, 5, 6
    w, am = p.w, "adaptation_module."
    nf = ns + na
    W1 = np.transpose(w[am + "tCNN.0.weight"], (0, 2, 1)).reshape(32, 5 * 32)     # [out, k*32 + in]: rows of the window
    W2 = np.transpose(w[am + "tCNN.1.weight"], (0, 2, 1)).reshape(16, 5 * 32)
    Wo = w[am + "outMLP.0._model.0.weight"]                                         # columns: channel-major flatten [16][10]
    Wo = np.transpose(Wo.reshape(Wo.shape[0], 16, 10), (0, 2, 1)).reshape(Wo.shape[0], 160)   # -> position-major [10][16]
    y0 = np.zeros(nf, np.float32)
    for k in range(3):
        y0 = _f32_fc(w[am + "inMLP.%d._model.0.weight" % k], w[am + "inMLP.%d._model.0.bias" % k], y0, True)
    c1z = _f32_fc(W1, w[am + "tCNN.0.bias"], np.tile(y0, 5), False)
    c2z = _f32_fc(W2, w[am + "tCNN.1.bias"], np.tile(c1z, 5), False)
    ry, r1, r2 = p.ring(5, 32, 1, y0), p.ring(4, 32, 2, c1z), p.ring(9, 16, 2, c2z)
    p.copy_obs(0, ns, X, 0); p.copy_prev(na, X, ns)                                 # RMA_model.py:87-89: the newest (state, action) row
    p.ring_load(ry, (H, 0)); p.ring_load(r1, (C1, 0)); p.ring_load(r2, (C2, 0))
    p.fc(am + "inMLP.0", (X, 0), (A, 0), "tanh"); p.fc(am + "inMLP.1", (A, 0), (B, 0), "tanh")
    p.fc(am + "inMLP.2", (B, 0), (P, 0), "tanh")
    p.dense(W1, w[am + "tCNN.0.bias"], (H, 0), (C1, 4 * 32), None)                  # conv1 over the 5 rows BEFORE the newest one
    p.ring_push((P, 0), ry)
    p.dense(W2, w[am + "tCNN.1.bias"], (C1, 0), (C2, 9 * 16), None)
    p.ring_push((C1, 4 * 32), r1)
    p.dense(Wo, w[am + "outMLP.0._model.0.bias"], (C2, 0), (A, 0), "tanh")
    p.ring_push((C2, 9 * 16), r2)
    z = p.fc(am + "outMLP.1", (A, 0), (X, nf), None)                                # z_hat next to flat_in (:101-105)
    p.fc("_hidden_layers.0", (X, 0), (A, 0), "tanh")
    f = p.fc("_hidden_layers.1", (A, 0), (B, 0), "tanh")
    p.bn("_hidden_layers.2", B, 0)
    p.fc("_logits.0", (B, 0), (A, 0), "tanh")
    nl = p.fc("_logits.1", (A, 0), (P, 0), None)
    p.flags = L.POL_VALUE_ONLY
    p.fc("_value_branch.0", (B, 0), (A, 0), "tanh")
    p.fc("_value_branch.1", (A, 0), (A, 128), "tanh")
    p.fc("_value_branch.2", (A, 128), (X, 0), None)
    return dict(widths=[max(32, nf + z), 32, 256, max(f, 128), 160, 160, 160], logits=(P, 0, nl), value=(X, 0))


def _cnn_estimator(p, D, ns, npar, na):
    """CNNestimator, use_estimate=False (StateEstimatorLSTM.py:250-283; train_LSTM.py:51-60): ns = num_states = obs_dim = 23"""
    X, P, A, B = 0, 1, 2, 3
    p.copy_obs(0, ns - 4, X, 0); p.copy_prev(na, X, ns - 4); p.copy_obs(ns - 4, 4, X, ns - 4 + na)   # cat(flat_in, gt_pendulum_state)
    p.fc("_hidden.0", (X, 0), (A, 0), "tanh")
    p.fc("_hidden.1", (A, 0), (B, 0), "tanh")
    nl = p.fc("_logits.0", (B, 0), (P, 0), None)
    p.flags = L.POL_VALUE_ONLY
    p.fc("_value_branch.0", (B, 0), (A, 0), "tanh")
    p.fc("_value_branch.1", (A, 0), (A, 128), "tanh")
    p.fc("_value_branch.2", (A, 128), (X, 0), None)
    return dict(widths=[32, 16, 256, 128], logits=(P, 0, nl), value=(X, 0))


def _cnn_estimator_estimate(p, D, ns, npar, na):
    """CNNestimator, use_estimate=True: the last four observation entries (pendulum state) are replaced by the TimeCNN
    estimate from the 32-step (obs[:19], previous action) history -- evaluated incrementally like _rma_full_adapt"""
    X, P, A, B, H, C1, C2 = 0, 1, 2, 3, 4, 5, 6
    w, em = p.w, "estimation_module."
    nf = ns - 4 + na
    W1 = np.transpose(w[em + "tCNN.0.weight"], (0, 2, 1)).reshape(32, 5 * 32)
    W2 = np.transpose(w[em + "tCNN.1.weight"], (0, 2, 1)).reshape(16, 5 * 32)
    Wo = w[em + "outMLP.0._model.0.weight"]
    Wo = np.transpose(Wo.reshape(Wo.shape[0], 16, 10), (0, 2, 1)).reshape(Wo.shape[0], 160)
    y0 = np.zeros(nf, np.float32)
    for k in range(2):
        y0 = _f32_fc(w[em + "inMLP.%d._model.0.weight" % k], w[em + "inMLP.%d._model.0.bias" % k], y0, True)
    c1z = _f32_fc(W1, w[em + "tCNN.0.bias"], np.tile(y0, 5), False)
    c2z = _f32_fc(W2, w[em + "tCNN.1.bias"], np.tile(c1z, 5), False)
    ry, r1, r2 = p.ring(5, 32, 1, y0), p.ring(4, 32, 2, c1z), p.ring(9, 16, 2, c2z)
    p.copy_obs(0, ns - 4, X, 0); p.copy_prev(na, X, ns - 4)
    p.ring_load(ry, (H, 0)); p.ring_load(r1, (C1, 0)); p.ring_load(r2, (C2, 0))
    p.fc(em + "inMLP.0", (X, 0), (A, 0), "tanh")
    p.fc(em + "inMLP.1", (A, 0), (P, 0), "tanh")
    p.dense(W1, w[em + "tCNN.0.bias"], (H, 0), (C1, 4 * 32), None)
    p.ring_push((P, 0), ry)
    p.dense(W2, w[em + "tCNN.1.bias"], (C1, 0), (C2, 9 * 16), None)
    p.ring_push((C1, 4 * 32), r1)
    p.dense(Wo, w[em + "outMLP.0._model.0.bias"], (C2, 0), (A, 0), "tanh")
    p.ring_push((C2, 9 * 16), r2)
    p.fc(em + "outMLP.1", (A, 0), (X, nf), None)                                     # the estimate next to flat_in
    p.fc("_hidden.0", (X, 0), (A, 0), "tanh")
    p.fc("_hidden.1", (A, 0), (B, 0), "tanh")
    nl = p.fc("_logits.0", (B, 0), (P, 0), None)
    p.flags = L.POL_VALUE_ONLY
    p.fc("_value_branch.0", (B, 0), (A, 0), "tanh")
    p.fc("_value_branch.1", (A, 0), (A, 128), "tanh")
    p.fc("_value_branch.2", (A, 128), (X, 0), None)
    return dict(widths=[32, 32, 256, 128, 160, 160, 160], logits=(P, 0, nl), value=(X, 0))


def _custom_mlp(p, D, ns, npar, na):
    """CustomMLP (models/PPO/MLP/CustomMLP.py:17-98): BatchNorm -> 256 -> 128 -> 128 -> 96 -> BatchNorm trunk on cat(obs, prev_actions)"""
    X, P, A, B = 0, 1, 2, 3
    p.copy_obs(0, D, X, 0); p.copy_prev(na, X, D)
    p.bn("_hidden_layers.0", X, 0)
    p.fc("_hidden_layers.1", (X, 0), (A, 0), "tanh")
    p.fc("_hidden_layers.2", (A, 0), (B, 0), "tanh")
    p.fc("_hidden_layers.3", (B, 0), (A, 0), "tanh")
    p.fc("_hidden_layers.4", (A, 0), (B, 0), "tanh")
    p.bn("_hidden_layers.5", B, 0)
    p.fc("_logits.0", (B, 0), (A, 0), "tanh")
    p.fc("_logits.1", (A, 0), (A, 64), "tanh")
    nl = p.fc("_logits.2", (A, 64), (P, 0), None)
    p.flags = L.POL_VALUE_ONLY
    p.fc("_value_branch.0", (B, 0), (A, 0), "tanh")
    p.fc("_value_branch.1", (A, 0), (A, 128), "tanh")
    p.fc("_value_branch.2", (A, 128), (X, 0), None)
    return dict(widths=[max(32, D + na), 16, 256, 128], logits=(P, 0, nl), value=(X, 0))


def _lstm_estimator(p, D, ns, npar, na, use_estimate):
    """LSTMestimator (StateEstimatorLSTM.py:15-147): observation = 15 drone values + 4 pendulum values (D = 19); the policy sees
    cat(o_t[:15], a_{t-1}, pendulum) where pendulum is the ground truth (use_estimate=False: no recurrence on the action path) or
    LSTMestimatorModule2's estimate from (o_{t-1}[:15], o_t[:15], a_{t-1}) (:174-197: MLP1 -> nn.LSTM(32, 32) -> MLP2(f + y)).
    h, c and the previous observation travel between steps in one-row rings (zeros at an episode start, as get_initial_state
    and RLlib's zero padding give)."""
    X, P, A, B, IN, S, G = 0, 1, 2, 3, 4, 5, 6
    w, em, no, H = p.w, "estimation_module.", D - 4, 32
    p.copy_obs(0, no, X, 0); p.copy_prev(na, X, no)
    if not use_estimate:
        p.copy_obs(no, 4, X, no + na)
        widths = [32, 16, 256, 128]
    else:
        r_o, r_h, r_c = p.ring(1, no, 1, np.zeros(no)), p.ring(1, H, 1, np.zeros(H)), p.ring(1, H, 1, np.zeros(H))
        p.ring_load(r_o, (IN, 0)); p.ring_load(r_h, (S, H)); p.ring_load(r_c, (S, 2 * H))
        p.copy_obs(0, no, IN, no); p.copy_prev(na, IN, 2 * no)                      # :77-78 cat(prev_o[:, :, :15].flatten(1), prev_a)
        p.fc(em + "MLP1.0", (IN, 0), (A, 0), "tanh")
        p.fc(em + "MLP1.1", (A, 0), (S, 0), "tanh")                                  # y
        Wg = np.concatenate([w[em + "LSTM.weight_ih_l0"], w[em + "LSTM.weight_hh_l0"]], axis=1)      # gates over [y | h]
        p.dense(Wg, w[em + "LSTM.bias_ih_l0"] + w[em + "LSTM.bias_hh_l0"], (S, 0), (G, 0), None)
        p.ring_push((IN, no), r_o)                                                  # o_t is the next step's o_{t-1}
        p.lstm_cell((G, 0), (S, H), H)
        p.ring_push((S, H), r_h); p.ring_push((S, 2 * H), r_c)
        W2 = w[em + "MLP2.0._model.0.weight"]
        p.dense(np.concatenate([W2, W2], axis=1), w[em + "MLP2.0._model.0.bias"], (S, 0), (A, 0), "tanh")   # MLP2(f + y), f = h'
        p.fc(em + "MLP2.1", (A, 0), (X, no + na), None)
        widths = [32, 16, 256, 128, 48, 96, 128]
    p.fc("_hidden.0", (X, 0), (A, 0), "tanh")
    p.fc("_hidden.1", (A, 0), (B, 0), "tanh")
    nl = p.fc("_logits.0", (B, 0), (P, 0), None)
    p.flags = L.POL_VALUE_ONLY
    p.fc("_value_branch.0", (B, 0), (A, 0), "tanh")
    p.fc("_value_branch.1", (A, 0), (A, 128), "tanh")
    p.fc("_value_branch.2", (A, 128), (X, 0), None)
    return dict(widths=widths, logits=(P, 0, nl), value=(X, 0))


def _custom_lstm(p, D, ns, npar, na):
    """CustomLSTM (models/PPO/CustomLSTM/CustomLSTM.py:14-105): features = BatchNorm(MLP1(cat(obs, prev_action))) [64],
    logits = _logits(nn.LSTM(64, 64)(features) + features), value from the features; h and c in one-row rings"""
    X, P, A, B, S, G = 0, 1, 2, 3, 4, 5
    w, H = p.w, 64
    r_h, r_c = p.ring(1, H, 1, np.zeros(H)), p.ring(1, H, 1, np.zeros(H))
    p.copy_obs(0, D, X, 0); p.copy_prev(na, X, D)
    p.ring_load(r_h, (S, H)); p.ring_load(r_c, (S, 2 * H))
    p.fc("MLP1.0", (X, 0), (S, 0), "tanh")
    p.bn("bn", S, 0)                                                                 # :83 BatchNorm1d over the feature axis
    Wg = np.concatenate([w["LSTM.weight_ih_l0"], w["LSTM.weight_hh_l0"]], axis=1)
    p.dense(Wg, w["LSTM.bias_ih_l0"] + w["LSTM.bias_hh_l0"], (S, 0), (G, 0), None)
    p.lstm_cell((G, 0), (S, H), H)
    p.ring_push((S, H), r_h); p.ring_push((S, 2 * H), r_c)
    Wl = w["_logits.0._model.0.weight"]
    nl = p.dense(np.concatenate([Wl, Wl], axis=1), w["_logits.0._model.0.bias"], (S, 0), (P, 0), None)   # _logits(f + features), f = h'
    p.flags = L.POL_VALUE_ONLY
    p.fc("_value_branch.0", (S, 0), (A, 0), "tanh")
    p.fc("_value_branch.1", (A, 0), (X, 0), None)
    return dict(widths=[max(32, D + na), 16, 128, 16, 3 * H, 4 * H], logits=(P, 0, nl), value=(X, 0))


def _rma_model_small(p, D, ns, npar, na, wide):
    """RMA_model_smaller (RMA_model.py:311-347; wide=False) and RMA_model_smaller2 (the definition Python keeps, :398-437; wide=True):
    RMA_model.forward (:262-292) over a 256-128 / 512-256 trunk.  ResBlock(x) = hidden(x) + x (:350-357) costs no extra op: the
    block's output is written right behind its input and the layer after it reads both halves through [W | W]."""
    X, P, A, B = 0, 1, 2, 3
    w = p.w
    p.copy_obs(0, ns, X, 0); p.copy_prev(na, X, ns); p.copy_obs(ns, npar, P, 0)
    p.fc("param_encoder.0", (P, 0), (A, 0), "tanh")
    z = p.fc("param_encoder.1", (A, 0), (X, ns + na), "tanh")
    p.fc("_hidden_layers.0", (X, 0), (A, 0), "tanh")
    f = p.fc("_hidden_layers.1", (A, 0), (B, 0), "tanh")
    p.bn("_hidden_layers.2", B, 0)
    if not wide:
        p.fc("_logits.0", (B, 0), (A, 0), "tanh")
        nl = p.fc("_logits.1", (A, 0), (P, 0), None)
        p.flags = L.POL_VALUE_ONLY
        p.fc("_value_branch.0", (B, 0), (A, 0), "tanh")
        p.fc("_value_branch.1", (A, 0), (A, 128), "tanh")
        p.fc("_value_branch.2", (A, 128), (X, 0), None)
        widths = [max(32, ns + na + z), 16, 256, 128]
    else:
        nl = p.fc("_logits.0", (B, 0), (P, 0), None)
        p.flags = L.POL_VALUE_ONLY
        p.fc("_value_branch.0.hidden.0", (B, 0), (B, f), "tanh")                        # ResBlock(256, 1): [features | hidden(features)]
        W1 = w["_value_branch.1._model.0.weight"]
        v = p.dense(np.concatenate([W1, W1], axis=1), w["_value_branch.1._model.0.bias"], (B, 0), (A, 0), "tanh")
        p.fc("_value_branch.2.hidden.0", (A, 0), (A, 2 * v), "tanh")                    # ResBlock(128, 2): [v | hidden(v)]
        p.fc("_value_branch.2.hidden.1", (A, 2 * v), (A, v), "tanh")
        W3 = w["_value_branch.3._model.0.weight"]
        p.dense(np.concatenate([W3, W3], axis=1), w["_value_branch.3._model.0.bias"], (A, 0), (X, 0), None)
        widths = [max(32, ns + na + z), 16, 512, 2 * f]
    return dict(widths=widths, logits=(P, 0, nl), value=(X, 0), aux=(X, ns + na, z))


def _custom_lstm_bigger(p, D, ns, npar, na, common_f):
    """CustomLSTMbigger (CustomLSTM.py:107-201) / CustomLSTMbiggerCommonF (:204-299): y = BatchNorm(MLP1(cat(obs, prev_action))) over two
    layers, logits = _logits(nn.LSTM(64, 64)(y) + y); the value head reads y (bigger) or LSTM(y) + y (CommonF)"""
    X, P, A, B, S, G = 0, 1, 2, 3, 4, 5
    w, H = p.w, 64
    r_h, r_c = p.ring(1, H, 1, np.zeros(H)), p.ring(1, H, 1, np.zeros(H))
    p.copy_obs(0, D, X, 0); p.copy_prev(na, X, D)
    p.ring_load(r_h, (S, H)); p.ring_load(r_c, (S, 2 * H))
    p.fc("MLP1.0", (X, 0), (A, 0), "tanh")
    p.fc("MLP1.1", (A, 0), (S, 0), "tanh")
    p.bn("bn", S, 0)
    Wg = np.concatenate([w["LSTM.weight_ih_l0"], w["LSTM.weight_hh_l0"]], axis=1)
    p.dense(Wg, w["LSTM.bias_ih_l0"] + w["LSTM.bias_hh_l0"], (S, 0), (G, 0), None)
    p.lstm_cell((G, 0), (S, H), H)
    p.ring_push((S, H), r_h); p.ring_push((S, 2 * H), r_c)
    Wl = w["_logits.0._model.0.weight"]
    p.dense(np.concatenate([Wl, Wl], axis=1), w["_logits.0._model.0.bias"], (S, 0), (A, 0), "tanh")      # _logits(f + y), f = h'
    nl = p.fc("_logits.1", (A, 0), (P, 0), None)
    p.flags = L.POL_VALUE_ONLY
    if common_f:
        Wv = w["_value_branch.0._model.0.weight"]
        p.dense(np.concatenate([Wv, Wv], axis=1), w["_value_branch.0._model.0.bias"], (S, 0), (A, 0), "tanh")
    else:
        p.fc("_value_branch.0", (S, 0), (A, 0), "tanh")
    p.fc("_value_branch.1", (A, 0), (A, 128), "tanh")
    p.fc("_value_branch.2", (A, 128), (X, 0), None)
    return dict(widths=[max(32, D + na), 16, 256, 16, 3 * H, 4 * H], logits=(P, 0, nl), value=(X, 0))


def _block_diag(mats):
    rows, cols = sum(m.shape[0] for m in mats), sum(m.shape[1] for m in mats)
    out, r, c = np.zeros((rows, cols), dtype=np.float64), 0, 0
    for m in mats:
        out[r:r + m.shape[0], c:c + m.shape[1]] = m
        r, c = r + m.shape[0], c + m.shape[1]
    return out


def _dsn_lstm(p, D, ns, npar, na):
    """DSN_LSTM_model (models/PPO/DSN_LSTM/DSN_LSTM_model.py:20-160): obs[:12] viewed [4, 3] and split into x / y / z columns, one
    MLP + BatchNorm + nn.LSTM per axis (32 / 32 / 16 wide), mixer on cat(LSTM outputs + features, prev_actions), value on the features.
    The three per-axis networks are independent, so they run here as ONE network with block-structured weights: layer 1 picks its
    axis' columns out of obs[:12], layers 2-3 are block-diagonal, and the three LSTMs are one 80-wide LSTM whose W_ih / W_hh are
    block-diagonal per gate (the cell update is elementwise).  h | c travel in one 160-wide ring."""
    X, P, A, B, S, G = 0, 1, 2, 3, 4, 5
    w = p.w
    axes, Hs = ("x", "y", "z"), (32, 32, 16)
    H = sum(Hs)
    f64 = lambda k: np.asarray(w[k], np.float64)
    r_hc = p.ring(1, 2 * H, 1, np.zeros(2 * H))
    p.copy_obs(0, 12, X, 0); p.copy_prev(na, S, 0)
    p.ring_load(r_hc, (S, na + H))
    W1 = []
    for k, ax in enumerate(axes):                                                    # :121-124 xyz_obs[..., k] = obs[k::3][:4]
        Wk = f64(ax + "_hidden.0._model.0.weight")
        E = np.zeros((Wk.shape[0], 12)); E[:, k::3] = Wk
        W1.append(E)
    cat_b = lambda i: np.concatenate([f64("%s_hidden.%d._model.0.bias" % (ax, i)) for ax in axes])
    p.dense(np.concatenate(W1, axis=0), cat_b(0), (X, 0), (A, 0), "tanh")
    p.dense(_block_diag([f64(ax + "_hidden.1._model.0.weight") for ax in axes]), cat_b(1), (A, 0), (B, 0), "tanh")
    p.dense(_block_diag([f64(ax + "_hidden.2._model.0.weight") for ax in axes]), cat_b(2), (B, 0), (S, na), "tanh")
    p.bn(tuple("bn_" + ax for ax in axes), S, na)                                    # features = cat(x_f, y_f, z_f)
    Wg, bg = np.zeros((4 * H, 2 * H)), np.zeros(4 * H)
    for gate in range(4):                                                            # i, f, g, o
        o = 0
        for ax, h in zip(axes, Hs):
            rows = slice(gate * H + o, gate * H + o + h)
            Wg[rows, o:o + h] = f64("LSTM_%s.weight_ih_l0" % ax)[gate * h:(gate + 1) * h]
            Wg[rows, H + o:H + o + h] = f64("LSTM_%s.weight_hh_l0" % ax)[gate * h:(gate + 1) * h]
            bg[rows] = (f64("LSTM_%s.bias_ih_l0" % ax) + f64("LSTM_%s.bias_hh_l0" % ax))[gate * h:(gate + 1) * h]
            o += h
    p.dense(Wg, bg, (S, na), (G, 0), None)
    p.lstm_cell((G, 0), (S, na + H), H)
    p.ring_push((S, na + H), r_hc)
    Wm = f64("mixer.0._model.0.weight")                                              # :137-138 cat(f + features, actions)
    p.dense(np.concatenate([Wm[:, H:H + na], Wm[:, :H], Wm[:, :H]], axis=1), f64("mixer.0._model.0.bias"), (S, 0), (A, 0), "tanh")
    nl = p.fc("mixer.1", (A, 0), (P, 0), None)
    p.flags = L.POL_VALUE_ONLY
    p.fc("_value_branch.0", (S, na), (A, 0), "tanh")
    p.fc("_value_branch.1", (A, 0), (A, 128), "tanh")
    p.fc("_value_branch.2", (A, 128), (X, 0), None)
    return dict(widths=[16, 16, 256, 160, na + 3 * H + 12, 4 * H], logits=(P, 0, nl), value=(X, 0))


_FAMILIES = {"CustomLSTM": _custom_lstm,
             "CustomLSTMbigger": lambda p, D, ns, npar, na: _custom_lstm_bigger(p, D, ns, npar, na, False),
             "CustomLSTMbiggerCommonF": lambda p, D, ns, npar, na: _custom_lstm_bigger(p, D, ns, npar, na, True),
             "DSN_LSTM_model": _dsn_lstm,
             "RMA_model_smaller": lambda p, D, ns, npar, na: _rma_model_small(p, D, ns, npar, na, False),
             "RMA_model_smaller2": lambda p, D, ns, npar, na: _rma_model_small(p, D, ns, npar, na, True), "LSTMestimator": lambda p, D, ns, npar, na: _lstm_estimator(p, D, ns, npar, na, False),
             "LSTMestimator_estimate": lambda p, D, ns, npar, na: _lstm_estimator(p, D, ns, npar, na, True),
             "CustomMLP": _custom_mlp, "CNNestimator": _cnn_estimator, "CNNestimator_estimate": _cnn_estimator_estimate, "RMA_full": _rma_full, "RMA_model": _rma_model, "SimpleMLPmodel": _simple_mlp, "RMA_full_adapt": _rma_full_adapt}


def compile_program(family, weights, obs_dim=22, num_states=16, num_params=6, num_actions=4):
    """state dict -> (qd_policy_desc, qd_policy_op array, float32 weight blob); host only"""
    if not callable(family) and family not in _FAMILIES:
        raise ValueError("unknown policy family %r (have %s)" % (family, sorted(_FAMILIES)))
    prog = _Program(weights)
    build = family if callable(family) else _FAMILIES[family]   # a callable builds its own layer program (same signature)
    lay = build(prog, int(obs_dim), int(num_states), int(num_params), int(num_actions))
    d = L.QdPolicyDesc()
    d.n_ops, d.n_bufs = len(prog.ops), len(lay["widths"])
    for b, x in enumerate(lay["widths"]):
        d.buf_width[b] = int(x)
    d.n_rings = len(prog.rings)
    for r, (rows, width, period, fill_off) in enumerate(prog.rings):
        d.ring[r].rows, d.ring[r].width, d.ring[r].period, d.ring[r].fill_off = rows, width, period, fill_off
    d.obs_dim, d.act_dim = int(obs_dim), int(num_actions)
    d.logits_buf, d.logits_off, d.n_logits = lay["logits"]
    d.value_buf, d.value_off = lay.get("value", (-1, 0))
    if "aux" in lay:
        d.aux_buf, d.aux_off, d.aux_dim = lay["aux"]
    ops = (L.QdPolicyOp * len(prog.ops))(*prog.ops)
    return d, ops, np.concatenate(prog.blob).astype(np.float32)


def random_weights(family, seed=0, num_states=16, num_params=6, num_actions=4, param_embed_dim=8, num_outputs=8):
    """random-init state dict of one of the reference's networks (Xavier-normal weights as the reference initialises them,
    small random biases, non-trivial BatchNorm statistics): for benchmarks and smoke tests, where no checkpoint exists"""
    rng = np.random.default_rng(seed)
    w = {}

    def fc(name, i, o):
        w[name + "._model.0.weight"] = (rng.normal(size=(o, i)) * np.sqrt(2.0 / (i + o))).astype(np.float32)
        w[name + "._model.0.bias"] = (0.1 * rng.normal(size=o)).astype(np.float32)

    def bn(name, n):
        w[name + ".weight"] = rng.uniform(0.5, 1.5, n).astype(np.float32)
        w[name + ".bias"] = (0.1 * rng.normal(size=n)).astype(np.float32)
        w[name + ".running_mean"] = (0.2 * rng.normal(size=n)).astype(np.float32)
        w[name + ".running_var"] = rng.uniform(0.25, 1.75, n).astype(np.float32)

    def lstm(name, hidden):   # nn.LSTM(hidden, hidden): uniform(-1/sqrt(H), 1/sqrt(H)) like torch
        k = 1.0 / np.sqrt(hidden)
        for nm, shape in (("weight_ih_l0", (4 * hidden, hidden)), ("weight_hh_l0", (4 * hidden, hidden)), ("bias_ih_l0", (4 * hidden,)),
                          ("bias_hh_l0", (4 * hidden,))):
            w["%s.%s" % (name, nm)] = rng.uniform(-k, k, shape).astype(np.float32)

    h_in = num_states + num_actions + param_embed_dim
    if family in ("CNNestimator", "CNNestimator_estimate"):   # num_states = 23 (LocalFrameFullStateEnv)
        ns = 23
        fc("_hidden.0", ns + num_actions, 256); fc("_hidden.1", 256, 128); fc("_logits.0", 128, num_outputs)
        fc("_value_branch.0", 128, 128); fc("_value_branch.1", 128, 128); fc("_value_branch.2", 128, 1)
        if family == "CNNestimator_estimate":
            em = "estimation_module."
            fc(em + "inMLP.0", ns, 32); fc(em + "inMLP.1", 32, 32)
            for name, (o, i) in (("tCNN.0", (32, 32)), ("tCNN.1", (16, 32))):
                w[em + name + ".weight"] = (rng.normal(size=(o, i, 5)) / np.sqrt(5 * i)).astype(np.float32)
                w[em + name + ".bias"] = (0.1 * rng.normal(size=o)).astype(np.float32)
            fc(em + "outMLP.0", 160, 32); fc(em + "outMLP.1", 32, 4)
        return w
    if family == "RMA_full_adapt":   # RMA_full's layers + the adaptation module (TimeCNN2, adapt_seq_len 32)
        w = random_weights("RMA_full", seed, num_states, num_params, num_actions, param_embed_dim, num_outputs)
        am, nf = "adaptation_module.", num_states + num_actions
        fc(am + "inMLP.0", nf, 32); fc(am + "inMLP.1", 32, 32); fc(am + "inMLP.2", 32, 32)
        for name, (o, i) in (("tCNN.0", (32, 32)), ("tCNN.1", (16, 32))):
            w[am + name + ".weight"] = (rng.normal(size=(o, i, 5)) / np.sqrt(5 * i)).astype(np.float32)
            w[am + name + ".bias"] = (0.1 * rng.normal(size=o)).astype(np.float32)
        fc(am + "outMLP.0", 160, 64); fc(am + "outMLP.1", 64, param_embed_dim)
        return w
    if family == "RMA_full":
        fc("param_encoder.0", num_params, 32); fc("param_encoder.1", 32, param_embed_dim)
        fc("_hidden_layers.0", h_in, 256); fc("_hidden_layers.1", 256, 128); bn("_hidden_layers.2", 128)
        fc("_logits.0", 128, 128); fc("_logits.1", 128, num_outputs)
        fc("_value_branch.0", 128, 128); fc("_value_branch.1", 128, 128); fc("_value_branch.2", 128, 1)
    elif family == "RMA_model":
        fc("param_encoder.0", num_params, 32); fc("param_encoder.1", 32, param_embed_dim)
        for k, (i, o) in enumerate(((h_in, 256), (256, 128), (128, 128), (128, 96))):
            fc("_hidden_layers.%d" % k, i, o)
        bn("_hidden_layers.4", 96)
        fc("_logits.0", 96, 64); fc("_logits.1", 64, 64); fc("_logits.2", 64, num_outputs)
        fc("_value_branch.0", 96, 128); fc("_value_branch.1", 128, 128); fc("_value_branch.2", 128, 1)
    elif family == "CustomMLP":
        x = num_states + num_params + num_actions
        bn("_hidden_layers.0", x)
        for k, (i, o) in enumerate(((x, 256), (256, 128), (128, 128), (128, 96))):
            fc("_hidden_layers.%d" % (k + 1), i, o)
        bn("_hidden_layers.5", 96)
        fc("_logits.0", 96, 64); fc("_logits.1", 64, 64); fc("_logits.2", 64, num_outputs)
        fc("_value_branch.0", 96, 128); fc("_value_branch.1", 128, 128); fc("_value_branch.2", 128, 1)
    elif family in ("LSTMestimator", "LSTMestimator_estimate"):   # 19-value observation (15 drone + 4 pendulum)
        fc("_hidden.0", 23, 256); fc("_hidden.1", 256, 128); fc("_logits.0", 128, num_outputs)
        fc("_value_branch.0", 128, 128); fc("_value_branch.1", 128, 128); fc("_value_branch.2", 128, 1)
        em = "estimation_module."
        fc(em + "MLP1.0", 34, 32); fc(em + "MLP1.1", 32, 32); fc(em + "MLP2.0", 32, 32); fc(em + "MLP2.1", 32, 4)
        for nm, shape in (("weight_ih_l0", (128, 32)), ("weight_hh_l0", (128, 32)), ("bias_ih_l0", (128,)), ("bias_hh_l0", (128,))):
            w[em + "LSTM." + nm] = (rng.normal(size=shape) * (0.17 if len(shape) == 2 else 0.1)).astype(np.float32)
    elif family in ("RMA_model_smaller", "RMA_model_smaller2"):
        wide = family.endswith("2")
        a, b = (512, 256) if wide else (256, 128)
        fc("param_encoder.0", num_params, 32); fc("param_encoder.1", 32, param_embed_dim)
        fc("_hidden_layers.0", h_in, a); fc("_hidden_layers.1", a, b); bn("_hidden_layers.2", b)
        if wide:
            fc("_logits.0", b, num_outputs)
            fc("_value_branch.0.hidden.0", 256, 256); fc("_value_branch.1", 256, 128)
            fc("_value_branch.2.hidden.0", 128, 128); fc("_value_branch.2.hidden.1", 128, 128); fc("_value_branch.3", 128, 1)
        else:
            fc("_logits.0", 128, 128); fc("_logits.1", 128, num_outputs)
            fc("_value_branch.0", 128, 128); fc("_value_branch.1", 128, 128); fc("_value_branch.2", 128, 1)
    elif family in ("CustomLSTM", "CustomLSTMbigger", "CustomLSTMbiggerCommonF"):   # input = cat(obs, prev_actions)
        x = num_states + num_params + num_actions
        if family == "CustomLSTM":
            fc("MLP1.0", x, 64); fc("_logits.0", 64, num_outputs); fc("_value_branch.0", 64, 128); fc("_value_branch.1", 128, 1)
        else:
            fc("MLP1.0", x, 64); fc("MLP1.1", 64, 64); fc("_logits.0", 64, 64); fc("_logits.1", 64, num_outputs)
            fc("_value_branch.0", 64, 128); fc("_value_branch.1", 128, 128); fc("_value_branch.2", 128, 1)
        bn("bn", 64)
        lstm("LSTM", 64)
    elif family == "DSN_LSTM_model":
        for ax, (a, b) in (("x", (64, 32)), ("y", (64, 32)), ("z", (32, 16))):
            fc(ax + "_hidden.0", 4, a); fc(ax + "_hidden.1", a, a); fc(ax + "_hidden.2", a, b)
            bn("bn_" + ax, b)
            lstm("LSTM_" + ax, b)
        fc("mixer.0", 84, 64); fc("mixer.1", 64, num_outputs)
        fc("_value_branch.0", 80, 128); fc("_value_branch.1", 128, 128); fc("_value_branch.2", 128, 1)
    elif family == "SimpleMLPmodel":
        x = num_states + num_params + num_actions
        for trunk, tail in (("_logits", (64, 64, num_outputs)), ("_value_branch", (128, 128, 1))):
            bn(trunk + ".0", x)
            for k, (i, o) in enumerate(((x, 256), (256, 128), (128, 128), (128, 96))):
                fc("%s.%d" % (trunk, k + 1), i, o)
            bn(trunk + ".5", 96)
            fc(trunk + ".6", 96, tail[0]); fc(trunk + ".7", tail[0], tail[1]); fc(trunk + ".8", tail[1], tail[2])
    else:
        raise ValueError("unknown policy family %r" % (family,))
    return w


class DevicePolicy:
    """One of the reference's policy networks, resident on the GPU.

    policy = DevicePolicy("RMA_full", state_dict)            # keys as in the reference's checkpoints
    actions = policy.forward(obs, prev_actions)               # [N,4] CUDA tensor; deterministic (Beta mean) action
    """

    DISTS = {"MyBetaDist": L.DIST_BETA, "MySquashedGaussian": L.DIST_SQUASHED_GAUSSIAN}   # distributions.py

    def __init__(self, family, weights, obs_dim=22, num_states=16, num_params=6, num_actions=4, device="cuda:0", dist="MyBetaDist"):
        self.lib = L.lib()
        self.device = torch.device(device)
        self.family, self.obs_dim, self.act_dim = family, int(obs_dim), int(num_actions)
        if dist not in self.DISTS:
            raise ValueError("unknown action distribution %r (have %s)" % (dist, sorted(self.DISTS)))
        d, ops, blob = compile_program(family, weights, obs_dim, num_states, num_params, num_actions)
        d.dist = self.DISTS[dist]                              # 'custom_action_dist' of the training scripts
        self.n_logits = int(d.n_logits)
        self.aux_dim = int(d.aux_dim)
        nbytes = self.lib.qd_policy_packed_bytes(C.byref(d), ops)
        if nbytes == 0:
            raise ValueError("invalid policy program: " + L.last_error())
        with torch.cuda.device(self.device):
            self.packed = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            handle = C.c_void_p()
            L.check(self.lib.qd_policy_create(C.byref(d), ops, blob.ctypes.data_as(C.c_void_p), blob.size, _ptr(self.packed),
                                              nbytes, C.byref(handle)))
        self.handle = handle
        self.kernel = int(self.lib.qd_policy_kernel(handle))   # 0: generic interpreter, > 0: specialised
        self.has_history = int(d.n_rings) > 0
        self.state = None                                       # per-env history (models with a time window), see reset_state
        self.n_weights, self.n_ops = int(blob.size), int(d.n_ops)

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            self.lib.qd_policy_destroy(h)
            self.handle = None

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _f32(self, t, shape):
        t = torch.as_tensor(t, device=self.device)
        t = t.to(dtype=torch.float32).contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError("expected shape %s, got %s" % (tuple(shape), tuple(t.shape)))
        return t

    def embedding(self, obs, prev_actions=None):
        """policy.model.z after a forward pass on obs [N,D] (rollout.py:83): the RMA networks' parameter embedding, [N, 8]"""
        n = int(obs.shape[0])
        obs = self._f32(obs, (n, self.obs_dim))
        prev = self._f32(prev_actions, (n, self.act_dim)) if prev_actions is not None else None
        out = torch.empty((n, self.aux_dim), dtype=torch.float32, device=self.device)
        L.check(self.lib.qd_policy_aux(self.handle, n, _ptr(obs), _ptr(prev), None, _ptr(out), self._stream()))
        return out

    def reset_state(self, n, mask=None):
        """start new episodes: the per-env history of a windowed model (RMA_full_adapt) is set to what the reference's
        zero-padded window gives; all n envs, or those with mask != 0.  No-op for feed-forward policies."""
        if not self.has_history:
            return
        nbytes = int(self.lib.qd_policy_state_bytes(self.handle, int(n)))
        if self.state is None or self.state.numel() != nbytes:
            if mask is not None:
                raise ValueError("the history must first be initialised for all envs")
            self.state = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        m = mask.to(device=self.device, dtype=torch.uint8).contiguous() if mask is not None else None
        L.check(self.lib.qd_policy_reset_state(self.handle, _ptr(self.state), int(n), _ptr(m), self._stream()))

    def _state_for(self, n):
        if not self.has_history:
            return None
        if self.state is None or self.state.numel() != int(self.lib.qd_policy_state_bytes(self.handle, int(n))):
            self.reset_state(n)
        return self.state

    def forward(self, obs, prev_actions=None, prev_truncated=None, want_logits=False, want_value=False, out=None,
                explore=False, seed=0, counter=0, want_logp=False):
        """model.forward + the action RLlib takes from MyBetaDist [+ value_function] for obs [N,D]:
        actions [N,4] (, logp [N]) (, logits [N,8]) (, value [N]).  explore=False: deterministic_sample (the Beta mean);
        explore=True: a Beta draw from the Philox stream (seed, counter)."""
        n = int(obs.shape[0])
        obs = self._f32(obs, (n, self.obs_dim))
        prev = self._f32(prev_actions, (n, self.act_dim)) if prev_actions is not None else None
        tr = prev_truncated.to(device=self.device, dtype=torch.uint8).contiguous() if prev_truncated is not None else None
        kw = dict(dtype=torch.float32, device=self.device)
        actions = torch.empty((n, self.act_dim), **kw) if out is None else out
        logp = torch.empty((n,), **kw) if want_logp else None
        logits = torch.empty((n, self.n_logits), **kw) if want_logits else None
        value = torch.empty((n,), **kw) if want_value else None
        L.check(self.lib.qd_policy_act(self.handle, n, _ptr(obs), _ptr(prev), _ptr(tr), int(bool(explore)), int(seed) & (2 ** 64 - 1),
                                       int(counter) & 0xFFFFFFFF, _ptr(self._state_for(n)), _ptr(actions), _ptr(logp), _ptr(logits),
                                       _ptr(value), self._stream()))
        res = (actions,) + ((logp,) if want_logp else ()) + ((logits,) if want_logits else ()) + ((value,) if want_value else ())
        return res[0] if len(res) == 1 else res

    def rollout(self, env, T, obs0, prev_actions0=None, want_logits=False, want_value=False, explore=False, seed=0, counter0=0,
                want_logp=False):
        """T closed-loop steps policy -> vector_step on a DeviceEnv, enqueued by one call:
        dict(obs [T,N,D], actions [T,N,4], reward [T,N], truncated [T,N][, logp, logits, value]) -- a PPO sample batch"""
        T, n = int(T), env.n
        kw = dict(dtype=torch.float32, device=self.device)
        obs0 = self._f32(obs0, (n, env.D))
        prev = self._f32(prev_actions0, (n, 4)) if prev_actions0 is not None else None
        out = dict(obs=torch.empty((T, n, env.D), **kw), actions=torch.empty((T, n, 4), **kw),
                   reward=torch.empty((T, n), **kw), truncated=torch.empty((T, n), dtype=torch.uint8, device=self.device))
        if want_logp:
            out["logp"] = torch.empty((T, n), **kw)
        if want_logits:
            out["logits"] = torch.empty((T, n, self.n_logits), **kw)
        if want_value:
            out["value"] = torch.empty((T, n), **kw)
        L.check(self.lib.qd_rollout_policy(env.handle, self.handle, T, _ptr(obs0), _ptr(prev), int(bool(explore)),
                                           int(seed) & (2 ** 64 - 1), int(counter0) & 0xFFFFFFFF, _ptr(self._state_for(n)),
                                           _ptr(out["obs"]), _ptr(out["actions"]),
                                           _ptr(out["reward"]), _ptr(out["truncated"]), _ptr(out.get("logp")), _ptr(out.get("logits")),
                                           _ptr(out.get("value")), self._stream()))
        return out
