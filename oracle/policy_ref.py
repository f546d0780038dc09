"""CPU float64 restatement of the reference's policy forward passes (actor + value, eval mode) and of its Beta
action distribution.  TEST INFRASTRUCTURE ONLY, like the rest of oracle/: the product never imports it.

Follows  models/PPO/RMA/RMA_model.py:77-110 (RMA_full.forward with train_adaptation=False, the train_PPO.py:39-45
configuration), :262-292 (RMA_model.forward), models/PPO/SimpleMLP/SimpleMLP.py:72-98 (SimpleMLPmodel) and
distributions.py:8-26 (MyBetaDist).  Pinned by tests/golden/policy_vectors.npz, which tests/golden/make_policy_golden.py
produced by running those reference classes themselves (over functional stand-ins for ray's SlimFC / TorchModelV2,
ray being absent: the vectors pin the reference's wiring, see that script's header).

Weights are a dict keyed like the reference's checkpoints (`policy_state.pkl` -> 'weights'): SlimFC layers are
`<seq>.<i>._model.0.weight/bias`, BatchNorm1d layers `<seq>.<i>.weight/bias/running_mean/running_var`.
"""
import numpy as np

BN_EPS = 1e-5  # torch.nn.BatchNorm1d default


def _fc(w, prefix, x, act):
    y = x @ np.asarray(w[prefix + "._model.0.weight"], dtype=np.float64).T + np.asarray(w[prefix + "._model.0.bias"], dtype=np.float64)
    return np.tanh(y) if act == "tanh" else y


def _bn(w, prefix, x):
    g, b = np.asarray(w[prefix + ".weight"], np.float64), np.asarray(w[prefix + ".bias"], np.float64)
    m, v = np.asarray(w[prefix + ".running_mean"], np.float64), np.asarray(w[prefix + ".running_var"], np.float64)
    return (x - m) / np.sqrt(v + BN_EPS) * g + b


def _seq(w, name, x, acts):
    """nn.Sequential `name` of SlimFC layers (act 'tanh' / None) and BatchNorm1d layers ('bn')"""
    for i, a in enumerate(acts):
        x = _bn(w, "%s.%d" % (name, i), x) if a == "bn" else _fc(w, "%s.%d" % (name, i), x, a)
    return x


def rma_full(w, obs, prev_actions, num_states=16, num_params=6):
    """RMA_model.py:77-110 with train_adaptation=False: z = param_encoder(obs[:, -num_params:]),
    features = hidden(cat(obs[:, :num_states], prev_actions, z)); returns (logits, value)"""
    obs, prev = np.asarray(obs, np.float64), np.asarray(prev_actions, np.float64)
    z = _seq(w, "param_encoder", obs[:, -num_params:], ["tanh", None])
    feat = _seq(w, "_hidden_layers", np.concatenate([obs[:, :num_states], prev, z], axis=-1), ["tanh", "tanh", "bn"])
    return _seq(w, "_logits", feat, ["tanh", None]), _seq(w, "_value_branch", feat, ["tanh", "tanh", None])[:, 0]


def rma_model(w, obs, prev_actions, num_states=16, num_params=6):
    """RMA_model.py:262-292: the encoder ends in tanh here, four hidden layers, three logits layers"""
    obs, prev = np.asarray(obs, np.float64), np.asarray(prev_actions, np.float64)
    z = _seq(w, "param_encoder", obs[:, num_states:num_states + num_params], ["tanh", "tanh"])
    feat = _seq(w, "_hidden_layers", np.concatenate([obs[:, :num_states], prev, z], axis=-1), ["tanh"] * 4 + ["bn"])
    return _seq(w, "_logits", feat, ["tanh", "tanh", None]), _seq(w, "_value_branch", feat, ["tanh", "tanh", None])[:, 0]


def simple_mlp(w, obs, prev_actions):
    """SimpleMLP.py:72-98: separate actor / critic trunks on cat(obs, prev_actions), BatchNorm on the input too"""
    x = np.concatenate([np.asarray(obs, np.float64), np.asarray(prev_actions, np.float64)], axis=-1)
    acts = ["bn"] + ["tanh"] * 4 + ["bn", "tanh", "tanh", None]
    return _seq(w, "_logits", x, acts), _seq(w, "_value_branch", x, acts)[:, 0]


def time_cnn2(w, x, prefix="adaptation_module", in_layers=3):
    """TimeCNN2.forward (RMA_model.py:155-191; in_layers=3) / TimeCNN.forward (StateEstimatorLSTM.py:312-336; in_layers=2):
    x [N, L, F] -> inMLP per time step -> Conv1d(32,32,5,stride 2) -> Conv1d(32,16,5) (no activation between them) ->
    flatten (channel-major) -> outMLP"""
    x = np.asarray(x, np.float64)
    n, L, _ = x.shape
    y = _seq(w, prefix + ".inMLP", x.reshape(n * L, -1), ["tanh"] * in_layers).reshape(n, L, -1)   # [N, L, 32]

    def conv(y, W, b, stride):
        W, b = np.asarray(W, np.float64), np.asarray(b, np.float64)          # W [out, in, k]
        k = W.shape[2]
        pos = range(0, y.shape[1] - k + 1, stride)
        return np.stack([np.einsum("nki,oik->no", y[:, p:p + k, :], W) + b for p in pos], axis=1)   # [N, P, out]
    c1 = conv(y, w[prefix + ".tCNN.0.weight"], w[prefix + ".tCNN.0.bias"], 2)
    c2 = conv(c1, w[prefix + ".tCNN.1.weight"], w[prefix + ".tCNN.1.bias"], 1)
    flat = np.transpose(c2, (0, 2, 1)).reshape(n, -1)                       # torch flattens [N, C, P]
    return _seq(w, prefix + ".outMLP", flat, ["tanh", None])


def rma_full_adapt(w, obs_history, action_history, num_states=16):
    """RMA_model.py:77-110 with train_adaptation=True (train_RMA.py:39-45): obs_history [N, L, D] (zero rows before the
    episode start), action_history [N, L, 4] (the action BEFORE each observation); z_hat = adaptation_module(history)
    takes the place of the parameter encoding; returns (logits, value, z_hat)"""
    oh, ah = np.asarray(obs_history, np.float64), np.asarray(action_history, np.float64)
    s_a = np.concatenate([oh[:, :, :num_states], ah], axis=-1)
    z_hat = time_cnn2(w, s_a)
    feat = _seq(w, "_hidden_layers", np.concatenate([s_a[:, -1], z_hat], axis=-1), ["tanh", "tanh", "bn"])
    return _seq(w, "_logits", feat, ["tanh", None]), _seq(w, "_value_branch", feat, ["tanh", "tanh", None])[:, 0], z_hat


def cnn_estimator(w, obs, prev_actions, num_states=23):
    """CNNestimator.forward with use_estimate=False, train_estimator=False (StateEstimatorLSTM.py:250-283; the train_LSTM.py:51-60
    configuration): hidden(cat(obs[:, :num_states-4], prev_actions, obs[:, num_states-4:])); returns (logits, value)"""
    obs, prev = np.asarray(obs, np.float64), np.asarray(prev_actions, np.float64)
    x = np.concatenate([obs[:, :num_states - 4], prev, obs[:, num_states - 4:num_states]], axis=-1)
    feat = _seq(w, "_hidden", x, ["tanh", "tanh"])
    return _seq(w, "_logits", feat, [None]), _seq(w, "_value_branch", feat, ["tanh", "tanh", None])[:, 0]


def cnn_estimator_hist(w, obs_history, action_history, num_states=23):
    """the same with use_estimate=True: the pendulum state is replaced by TimeCNN(32-step history of (obs[:19], previous
    action)); returns (logits, value, estimate)"""
    oh, ah = np.asarray(obs_history, np.float64), np.asarray(action_history, np.float64)
    o_a = np.concatenate([oh[:, :, :num_states - 4], ah], axis=-1)
    est = time_cnn2(w, o_a, "estimation_module", in_layers=2)
    feat = _seq(w, "_hidden", np.concatenate([o_a[:, -1], est], axis=-1), ["tanh", "tanh"])
    return _seq(w, "_logits", feat, [None]), _seq(w, "_value_branch", feat, ["tanh", "tanh", None])[:, 0], est


def custom_mlp(w, obs, prev_actions):
    """CustomMLP.forward (models/PPO/MLP/CustomMLP.py:75-98): one trunk on cat(obs, prev_actions) with BatchNorm at both ends,
    actor and critic heads on its features"""
    x = np.concatenate([np.asarray(obs, np.float64), np.asarray(prev_actions, np.float64)], axis=-1)
    feat = _seq(w, "_hidden_layers", x, ["bn"] + ["tanh"] * 4 + ["bn"])
    return _seq(w, "_logits", feat, ["tanh", "tanh", None]), _seq(w, "_value_branch", feat, ["tanh", "tanh", None])[:, 0]


def lstm_estimator(w, obs_seq, action_seq, use_estimate=True):
    """LSTMestimator.forward_rnn (StateEstimatorLSTM.py:100-118) over whole episodes: obs_seq [B, T, 19], action_seq [B, T, 4]
    (the action taken AFTER each observation); LSTMestimatorModule2 (:174-197) = MLP1 -> nn.LSTM(32, 32) (gate order i, f, g, o)
    -> MLP2(f + y), zero initial state, zero o_{t-1} / a_{t-1} at the episode start; returns (logits [B,T,8], value [B,T],
    estimates [B,T,4])"""
    o, a = np.asarray(obs_seq, np.float64), np.asarray(action_seq, np.float64)
    Bn, Tn, _ = o.shape
    f64 = lambda k: np.asarray(w[k], np.float64)
    em = "estimation_module."
    Wih, Whh, b = f64(em + "LSTM.weight_ih_l0"), f64(em + "LSTM.weight_hh_l0"), f64(em + "LSTM.bias_ih_l0") + f64(em + "LSTM.bias_hh_l0")
    sig = lambda x: 1.0 / (1.0 + np.exp(-x))
    h, c = np.zeros((Bn, 32)), np.zeros((Bn, 32))
    logits, value, est = [], [], []
    for t in range(Tn):
        o_prev = o[:, t - 1, :15] if t > 0 else np.zeros((Bn, 15))
        a_prev = a[:, t - 1] if t > 0 else np.zeros((Bn, 4))
        x = np.concatenate([o_prev, o[:, t, :15], a_prev], axis=-1)
        y = _seq(w, em + "MLP1", x, ["tanh", "tanh"])
        g = y @ Wih.T + h @ Whh.T + b
        i_, f_, g_, o_ = sig(g[:, :32]), sig(g[:, 32:64]), np.tanh(g[:, 64:96]), sig(g[:, 96:])
        c = f_ * c + i_ * g_
        h = o_ * np.tanh(c)
        e = _seq(w, em + "MLP2", h + y, ["tanh", None])
        pend = e if use_estimate else o[:, t, 15:]
        feat = _seq(w, "_hidden", np.concatenate([x[:, -19:], pend], axis=-1), ["tanh", "tanh"])
        logits.append(_seq(w, "_logits", feat, [None])); value.append(_seq(w, "_value_branch", feat, ["tanh", "tanh", None])[:, 0]); est.append(e)
    return np.stack(logits, 1), np.stack(value, 1), np.stack(est, 1)


def custom_lstm(w, obs_seq, action_seq):
    """CustomLSTM.forward_rnn (CustomLSTM.py:80-87) over whole episodes: features = BatchNorm(MLP1(cat(obs, prev_action))),
    logits = _logits(LSTM(features) + features), value = _value_branch(features); returns (logits [B,T,8], value [B,T])"""
    o, a = np.asarray(obs_seq, np.float64), np.asarray(action_seq, np.float64)
    Bn, Tn, _ = o.shape
    f64 = lambda k: np.asarray(w[k], np.float64)
    Wih, Whh, b = f64("LSTM.weight_ih_l0"), f64("LSTM.weight_hh_l0"), f64("LSTM.bias_ih_l0") + f64("LSTM.bias_hh_l0")
    H = Whh.shape[1]
    sig = lambda x: 1.0 / (1.0 + np.exp(-x))
    h, c = np.zeros((Bn, H)), np.zeros((Bn, H))
    logits, value = [], []
    for t in range(Tn):
        a_prev = a[:, t - 1] if t > 0 else np.zeros((Bn, a.shape[2]))
        feat = _bn(w, "bn", _seq(w, "MLP1", np.concatenate([o[:, t], a_prev], axis=-1), ["tanh"]))
        g = feat @ Wih.T + h @ Whh.T + b
        c = sig(g[:, H:2 * H]) * c + sig(g[:, :H]) * np.tanh(g[:, 2 * H:3 * H])
        h = sig(g[:, 3 * H:]) * np.tanh(c)
        logits.append(_seq(w, "_logits", h + feat, [None])); value.append(_seq(w, "_value_branch", feat, ["tanh", None])[:, 0])
    return np.stack(logits, 1), np.stack(value, 1)


def rma_model_smaller(w, obs, prev_actions, num_states=16, num_params=6):
    """RMA_model_smaller (RMA_model.py:311-347): RMA_model.forward (:262-292) over RMA_full-sized layers, the encoder ending in tanh;
    returns (logits, value, z)"""
    obs, prev = np.asarray(obs, np.float64), np.asarray(prev_actions, np.float64)
    z = _seq(w, "param_encoder", obs[:, num_states:num_states + num_params], ["tanh", "tanh"])
    feat = _seq(w, "_hidden_layers", np.concatenate([obs[:, :num_states], prev, z], axis=-1), ["tanh", "tanh", "bn"])
    return _seq(w, "_logits", feat, ["tanh", None]), _seq(w, "_value_branch", feat, ["tanh", "tanh", None])[:, 0], z


def rma_model_smaller2(w, obs, prev_actions, num_states=16, num_params=6):
    """RMA_model_smaller2 (the definition Python keeps, RMA_model.py:398-437): 512 -> 256 trunk, a single linear logits layer, value
    head ResBlock(256, 1) -> 128 -> ResBlock(128, 2) -> 1 with ResBlock(x) = hidden(x) + x (:350-357); returns (logits, value, z)"""
    obs, prev = np.asarray(obs, np.float64), np.asarray(prev_actions, np.float64)
    z = _seq(w, "param_encoder", obs[:, num_states:num_states + num_params], ["tanh", "tanh"])
    feat = _seq(w, "_hidden_layers", np.concatenate([obs[:, :num_states], prev, z], axis=-1), ["tanh", "tanh", "bn"])
    v = _seq(w, "_value_branch.0.hidden", feat, ["tanh"]) + feat
    v = _fc(w, "_value_branch.1", v, "tanh")
    v = _seq(w, "_value_branch.2.hidden", v, ["tanh", "tanh"]) + v
    return _seq(w, "_logits", feat, [None]), _fc(w, "_value_branch.3", v, None)[:, 0], z


def _lstm_step(w, prefix, x, h, c):
    """one nn.LSTM step, gate order i, f, g, o"""
    f64 = lambda k: np.asarray(w[k], np.float64)
    H = h.shape[1]
    g = x @ f64(prefix + ".weight_ih_l0").T + h @ f64(prefix + ".weight_hh_l0").T + f64(prefix + ".bias_ih_l0") + f64(prefix + ".bias_hh_l0")
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    c = sig(g[:, H:2 * H]) * c + sig(g[:, :H]) * np.tanh(g[:, 2 * H:3 * H])
    return sig(g[:, 3 * H:]) * np.tanh(c), c


def custom_lstm_bigger(w, obs_seq, action_seq, common_f=False):
    """CustomLSTMbigger.forward_rnn (CustomLSTM.py:171-178): y = BatchNorm(MLP1(cat(obs, prev_action))) (two layers), logits =
    _logits(LSTM(y) + y), value = _value_branch(y); CustomLSTMbiggerCommonF (:268-276, common_f=True): the value head reads
    LSTM(y) + y too; returns (logits [B,T,8], value [B,T])"""
    o, a = np.asarray(obs_seq, np.float64), np.asarray(action_seq, np.float64)
    Bn, Tn, _ = o.shape
    H = np.asarray(w["LSTM.weight_hh_l0"]).shape[1]
    h, c = np.zeros((Bn, H)), np.zeros((Bn, H))
    logits, value = [], []
    for t in range(Tn):
        a_prev = a[:, t - 1] if t > 0 else np.zeros((Bn, a.shape[2]))
        y = _bn(w, "bn", _seq(w, "MLP1", np.concatenate([o[:, t], a_prev], axis=-1), ["tanh", "tanh"]))
        h, c = _lstm_step(w, "LSTM", y, h, c)
        logits.append(_seq(w, "_logits", h + y, ["tanh", None]))
        value.append(_seq(w, "_value_branch", h + y if common_f else y, ["tanh", "tanh", None])[:, 0])
    return np.stack(logits, 1), np.stack(value, 1)


def dsn_lstm(w, obs_seq, action_seq):
    """DSN_LSTM_model.forward_rnn (DSN_LSTM_model.py:119-141): obs[:12] viewed [4, 3] and split into its x / y / z columns, one
    MLP + BatchNorm + nn.LSTM per axis (32, 32, 16 wide), mixer on cat(LSTM outputs + features, prev_actions), value head on the
    features; returns (logits [B,T,8], value [B,T])"""
    o, a = np.asarray(obs_seq, np.float64), np.asarray(action_seq, np.float64)
    Bn, Tn, _ = o.shape
    axes = (("x", 32), ("y", 32), ("z", 16))
    st = {ax: (np.zeros((Bn, H)), np.zeros((Bn, H))) for ax, H in axes}
    logits, value = [], []
    for t in range(Tn):
        a_prev = a[:, t - 1] if t > 0 else np.zeros((Bn, a.shape[2]))
        xyz = o[:, t, :12].reshape(Bn, 4, 3)
        feats, outs = [], []
        for k, (ax, H) in enumerate(axes):
            f = _bn(w, "bn_" + ax, _seq(w, ax + "_hidden", xyz[:, :, k], ["tanh", "tanh", "tanh"]))
            st[ax] = _lstm_step(w, "LSTM_" + ax, f, *st[ax])
            feats.append(f); outs.append(st[ax][0])
        feat = np.concatenate(feats, axis=-1)
        logits.append(_seq(w, "mixer", np.concatenate([np.concatenate(outs, axis=-1) + feat, a_prev], axis=-1), ["tanh", None]))
        value.append(_seq(w, "_value_branch", feat, ["tanh", "tanh", None])[:, 0])
    return np.stack(logits, 1), np.stack(value, 1)


FAMILIES = {"rma_full": rma_full, "rma_model": rma_model, "simple_mlp": simple_mlp, "custom_mlp": custom_mlp}


def beta_params(logits):
    """distributions.py:8-17: clamp to +-50, softplus + 1, first half = alpha (concentration1), second = beta"""
    x = np.clip(np.asarray(logits, np.float64), -50, 50)
    x = np.log(np.exp(x) + 1.0) + 1.0
    h = x.shape[-1] // 2
    return x[..., :h], x[..., h:]


def beta_mean_action(logits):
    """MyBetaDist.deterministic_sample (distributions.py:24-26): the Beta mean, no squashing"""
    a, b = beta_params(logits)
    return a / (a + b)


def beta_logp(logits, x):
    """MyBetaDist.logp (distributions.py:19-22): x clamped to [0.01, 0.99], summed over action dimensions"""
    from scipy.special import gammaln
    a, b = beta_params(logits)
    x = np.clip(np.asarray(x, np.float64), 1e-2, 1 - 1e-2)
    lp = (a - 1) * np.log(x) + (b - 1) * np.log1p(-x) - (gammaln(a) + gammaln(b) - gammaln(a + b))
    return lp.sum(-1)


def squashed_gaussian_mean_action(logits):
    """MySquashedGaussian.deterministic_sample (distributions.py:64-66, :103-106): sigmoid of the mean, clamped to [0, 1]"""
    x = np.asarray(logits, np.float64)
    mean = x[..., :x.shape[-1] // 2]
    return np.clip(1.0 / (1.0 + np.exp(-mean)), 0.0, 1.0)


def squashed_gaussian_logp(logits, x):
    """MySquashedGaussian.logp (distributions.py:73-85 with _unsquash :108-112).  QUIRK: the class squashes with a sigmoid but
    un-squashes with atanh(2 x - 1), which is half the pre-squash value; restated as written"""
    lg = np.asarray(logits, np.float64)
    h = lg.shape[-1] // 2
    mean, log_std = lg[..., :h], np.clip(lg[..., h:], -5, 5)
    std = np.exp(log_std)
    th = np.clip(np.asarray(x, np.float64) * 2.0 - 1.0, -1.0 + 1e-4, 1.0 - 1e-4)
    u = np.arctanh(th)
    lp = -0.5 * ((u - mean) / std) ** 2 - log_std - 0.5 * np.log(2 * np.pi)
    return np.clip(lp, -100, 100).sum(-1) - np.log(1 - th ** 2 + 1e-4).sum(-1)
