"""The policy oracle (oracle/policy_ref.py) against the outputs of the reference's own model classes
(tests/golden/policy_vectors.npz, see tests/golden/make_policy_golden.py)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def PG():
    return np.load(os.path.join(HERE, "golden", "policy_vectors.npz"))


def weights_of(PG, tag):
    return {k: PG[tag + "/" + k] for k in PG[tag + "_keys"]}


@pytest.mark.parametrize("tag", ["rma_full", "rma_model", "simple_mlp", "custom_mlp"])
def test_policy_oracle_vs_reference_models(PG, tag):
    from oracle import policy_ref as P
    w = weights_of(PG, tag)
    logits, value = P.FAMILIES[tag](w, PG["obs"], PG["prev_actions"])
    np.testing.assert_allclose(logits, PG[tag + "_logits"], atol=3e-6)       # the reference runs in float32
    np.testing.assert_allclose(value, PG[tag + "_value"], atol=3e-6)
    np.testing.assert_allclose(P.beta_mean_action(PG[tag + "_logits"]), PG[tag + "_action"], atol=1e-6)
    np.testing.assert_allclose(P.beta_logp(PG[tag + "_logits"], PG[tag + "_action"]), PG[tag + "_logp"], atol=2e-5)


FAMILY_OF = {"rma_full": "RMA_full", "rma_model": "RMA_model", "simple_mlp": "SimpleMLPmodel", "custom_mlp": "CustomMLP"}


@pytest.mark.parametrize("tag", ["rma_full", "rma_model", "simple_mlp", "custom_mlp"])
def test_policy_programs_compile_on_the_host(PG, tag):
    """state dict -> layer program -> qd_policy_packed_bytes (host-only entry point of the C ABI): the program passes
    the library's validation and the packed blob has the size the padded tilings imply"""
    import ctypes as C
    from mujoco_drone_amd import _lib as L
    from mujoco_drone_amd.policy import compile_program
    d, ops, blob = compile_program(FAMILY_OF[tag], weights_of(PG, tag))
    nbytes = L.lib().qd_policy_packed_bytes(C.byref(d), ops)
    assert nbytes > 0, L.last_error()
    # [program ints | small floats (biases padded to 16-wide tiles, affine scale / shift) | packed float32 weights (the interpreter's) |
    #  the same weights as float16 pairs in 16 x 32 tiles (the specialised kernels')]
    prog = 16 * len(ops) + 4 * 32          # op descriptors + one sentinel step per wave
    small = weights = 0
    for op in ops:
        if op.kind == L.POL_DENSE:
            k16, nt = (op.in_dim + 15) // 16, (op.out_dim + 15) // 16
            weights += nt * k16 * 256 + nt * ((op.in_dim + 31) // 32) * 512
            small += nt * 16
            for w in range(4):                                            # per-wave step lists
                slots = (nt - w + 3) // 4 if w < nt else 0
                while slots > 0:
                    u = 4 if slots >= 4 else 2 if slots >= 2 else 1
                    prog += 32 * ((k16 + 3) // 4)
                    slots -= u
        elif op.kind == L.POL_AFFINE:
            small += (2 * op.out_dim + 3) // 4 * 4
    assert nbytes == 4 * (prog + small + weights)
    assert d.n_logits == 8 and blob.dtype == np.float32
    # a program that reads outside its buffer is refused with a message, not run
    bad = type(ops)(*ops)
    bad[3].in_dim = 5000
    assert L.lib().qd_policy_packed_bytes(C.byref(d), bad) == 0 and "op 3" in L.last_error()


def test_adaptation_oracle_vs_reference_model(PG):
    """RMA_full with train_adaptation=True (TimeCNN2 over the 32-step history), incl. zero-padded young episodes"""
    from oracle import policy_ref as P
    w = weights_of(PG, "rma_adapt")
    logits, value, z_hat = P.rma_full_adapt(w, PG["rma_adapt_obs_history"], PG["rma_adapt_action_history"])
    np.testing.assert_allclose(z_hat, PG["rma_adapt_z_hat"], atol=3e-6)
    np.testing.assert_allclose(logits, PG["rma_adapt_logits"], atol=3e-6)
    np.testing.assert_allclose(value, PG["rma_adapt_value"], atol=3e-6)


def test_cnn_estimator_oracle_vs_reference_model(PG):
    """train_LSTM.py's network (CNNestimator), feed-forward and with the TimeCNN estimate in the loop"""
    from oracle import policy_ref as P
    logits, value = P.cnn_estimator(weights_of(PG, "cnn_est_ff"), PG["obs23"], PG["prev_actions"])
    np.testing.assert_allclose(logits, PG["cnn_est_ff_logits"], atol=3e-6)
    np.testing.assert_allclose(value, PG["cnn_est_ff_value"], atol=3e-6)
    logits, value, est = P.cnn_estimator_hist(weights_of(PG, "cnn_est_hist"), PG["cnn_est_hist_obs_history"], PG["cnn_est_hist_action_history"])
    np.testing.assert_allclose(est, PG["cnn_est_hist_estimate"], atol=3e-6)
    np.testing.assert_allclose(logits, PG["cnn_est_hist_logits"], atol=3e-6)
    np.testing.assert_allclose(value, PG["cnn_est_hist_value"], atol=3e-6)


def test_lstm_estimator_oracle_vs_reference_model(PG):
    """LSTMestimator.forward_rnn with the nn.LSTM estimate in the loop, 24-step episodes from the zero initial state"""
    from oracle import policy_ref as P
    logits, value, est = P.lstm_estimator(weights_of(PG, "lstm_est"), PG["lstm_est_obs_seq"], PG["lstm_est_action_seq"])
    np.testing.assert_allclose(est, PG["lstm_est_estimates"], atol=3e-6)
    np.testing.assert_allclose(logits, PG["lstm_est_logits"], atol=3e-6)
    np.testing.assert_allclose(value, PG["lstm_est_value"], atol=3e-6)


def test_custom_lstm_oracle_vs_reference_model(PG):
    """CustomLSTM.forward_rnn: the LSTM in the action path, BatchNorm on the features, 24-step episodes"""
    from oracle import policy_ref as P
    logits, value = P.custom_lstm(weights_of(PG, "custom_lstm"), PG["custom_lstm_obs_seq"], PG["custom_lstm_action_seq"])
    np.testing.assert_allclose(logits, PG["custom_lstm_logits"], atol=3e-6)
    np.testing.assert_allclose(value, PG["custom_lstm_value"], atol=3e-6)


def test_smaller_rma_variants_oracle_vs_reference_models(PG):
    """RMA_model_smaller and RMA_model_smaller2 (residual blocks in the value head), the variants train_PPO.py / evaluation.py import"""
    from oracle import policy_ref as P
    for tag, fn in (("rma_smaller", P.rma_model_smaller), ("rma_smaller2", P.rma_model_smaller2)):
        logits, value, z = fn(weights_of(PG, tag), PG["obs"], PG["prev_actions"])
        np.testing.assert_allclose(logits, PG[tag + "_logits"], atol=3e-6, err_msg=tag)
        np.testing.assert_allclose(value, PG[tag + "_value"], atol=3e-6, err_msg=tag)
        np.testing.assert_allclose(z, PG[tag + "_z"], atol=3e-6, err_msg=tag)


def test_recurrent_variants_oracle_vs_reference_models(PG):
    """CustomLSTMbigger, CustomLSTMbiggerCommonF and DSN_LSTM_model: forward_rnn over 24-step episodes from the zero state"""
    from oracle import policy_ref as P
    for tag, common in (("lstm_bigger", False), ("lstm_common_f", True)):
        logits, value = P.custom_lstm_bigger(weights_of(PG, tag), PG[tag + "_obs_seq"], PG[tag + "_action_seq"], common)
        np.testing.assert_allclose(logits, PG[tag + "_logits"], atol=3e-6, err_msg=tag)
        np.testing.assert_allclose(value, PG[tag + "_value"], atol=3e-6, err_msg=tag)
    logits, value = P.dsn_lstm(weights_of(PG, "dsn_lstm"), PG["dsn_lstm_obs_seq"], PG["dsn_lstm_action_seq"])
    np.testing.assert_allclose(logits, PG["dsn_lstm_logits"], atol=3e-6)
    np.testing.assert_allclose(value, PG["dsn_lstm_value"], atol=3e-6)


@pytest.mark.parametrize("tag,family", [("rma_adapt", "RMA_full_adapt"), ("cnn_est_ff", "CNNestimator"), ("cnn_est_hist", "CNNestimator_estimate"),
                                        ("lstm_est", "LSTMestimator_estimate"), ("custom_lstm", "CustomLSTM"), ("rma_smaller", "RMA_model_smaller"),
                                        ("rma_smaller2", "RMA_model_smaller2"), ("lstm_bigger", "CustomLSTMbigger"),
                                        ("lstm_common_f", "CustomLSTMbiggerCommonF"), ("dsn_lstm", "DSN_LSTM_model")])
def test_every_family_program_passes_host_validation(PG, tag, family):
    """each remaining family's layer program is accepted by the library's host-side validation (qd_policy_packed_bytes)"""
    import ctypes as C
    from mujoco_drone_amd import _lib as L
    from mujoco_drone_amd.policy import compile_program
    dims = {"cnn_est_ff": dict(obs_dim=23, num_states=23), "cnn_est_hist": dict(obs_dim=23, num_states=23),
            "lstm_est": dict(obs_dim=19, num_states=19)}.get(tag, {})
    d, ops, blob = compile_program(family, weights_of(PG, tag), **dims)
    assert L.lib().qd_policy_packed_bytes(C.byref(d), ops) > 0, L.last_error()
    assert d.n_logits == 8 and d.n_ops <= 32


def test_squashed_gaussian_oracle_vs_reference_distribution(PG):
    """MySquashedGaussian (the distribution the scripts import next to MyBetaDist): deterministic action and logp, incl. log_std
    at the clamp and actions at the edge of the squashing range"""
    from oracle import policy_ref as P
    np.testing.assert_allclose(P.squashed_gaussian_mean_action(PG["sg_logits"]), PG["sg_action"], atol=2e-7)
    for x, want in ((PG["sg_action"], PG["sg_logp_action"]), (PG["sg_x"], PG["sg_logp_x"])):
        np.testing.assert_allclose(P.squashed_gaussian_logp(PG["sg_logits"], x), want, rtol=5e-5, atol=3e-5)   # float32 atanh at the clamp


def test_constexpr_network_tables_match_the_layer_programs(monkeypatch, capsys):
    """the compile-time specialisations of csrc/qd_policy_static.h are printed from the Python layer programs
    (tools/emit_policy_arch.py); the header must contain exactly what the emitter prints today, or the library would silently
    fall back to the interpreter for that network"""
    import importlib.util
    import sys
    root = os.path.dirname(HERE)
    spec = importlib.util.spec_from_file_location("emit_policy_arch", os.path.join(root, "tools", "emit_policy_arch.py"))
    emit = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(emit)
    header = open(os.path.join(root, "mujoco-drone_amd", "csrc", "qd_policy_static.h")).read()
    for family, arch in (("RMA_model_smaller", "ArchRmaSmaller"), ("RMA_model_smaller2", "ArchRmaSmaller2"), ("CustomLSTM", "ArchCustomLstm"),
                         ("CustomLSTMbigger", "ArchLstmBigger"), ("CustomLSTMbiggerCommonF", "ArchLstmCommonF"), ("DSN_LSTM_model", "ArchDsnLstm")):
        monkeypatch.setattr(sys, "argv", ["emit_policy_arch.py", family, arch])
        emit.main()
        out = capsys.readouterr().out
        assert out.strip() in header, family
