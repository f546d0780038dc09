"""The train-batch statistics oracle (oracle/stats_ref.py) against the numbers the reference's own callback logs
(tests/golden/stats_vectors.npz, made by tests/golden/make_stats_golden.py running MyCallbacks.on_learn_on_batch)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def SG():
    return np.load(os.path.join(HERE, "golden", "stats_vectors.npz"))


@pytest.mark.parametrize("tag", ["small", "batch"])
def test_column_stats_oracle_vs_reference_callback(SG, tag):
    from oracle import stats_ref as S
    for key, what in (("obs", "obs"), ("actions", "act")):
        got = S.column_stats(SG["%s_%s" % (tag, key)])
        np.testing.assert_array_equal(got["min"], SG["%s_min_%s" % (tag, what)])            # exact: min / max pick elements
        np.testing.assert_array_equal(got["max"], SG["%s_max_%s" % (tag, what)])
        np.testing.assert_allclose(got["mean"], SG["%s_mean_%s" % (tag, what)], rtol=2e-6, atol=1e-6)   # the reference sums in float32
        np.testing.assert_allclose(got["var"], SG["%s_var_%s" % (tag, what)], rtol=2e-5, atol=1e-6)


def test_episode_stats_oracle_bookkeeping():
    """an episode ends AT the truncated step (its reward counts); running episodes carry over between fragments"""
    from oracle import stats_ref as S
    reward = np.array([[1.0, 10.0], [2.0, 20.0], [3.0, 30.0], [4.0, 40.0]])
    trunc = np.array([[0, 0], [1, 0], [0, 0], [0, 1]])
    rets, lens, carry = S.episode_stats(reward, trunc)
    assert sorted(rets.tolist()) == [3.0, 100.0] and sorted(lens.tolist()) == [2.0, 4.0]
    np.testing.assert_array_equal(carry, [[7.0, 2.0], [0.0, 0.0]])
    rets2, lens2, carry2 = S.episode_stats(reward[:2], trunc[:2], carry)
    assert rets2.tolist() == [10.0] and lens2.tolist() == [4.0]                            # 3 + 4 + 1 + 2 over 4 steps
    np.testing.assert_array_equal(carry2, [[0.0, 0.0], [30.0, 2.0]])


def test_stats_entry_points_validate_on_the_host():
    """host-only behaviour of the C ABI: workspace sizes, refusal of unsupported widths / empty batches with a message"""
    import ctypes as C
    from mujoco_drone_amd import _lib as L
    lib = L.lib()
    assert lib.qd_column_stats_workspace_bytes(22) == 1024 * 22 * 4 * 8 and lib.qd_column_stats_workspace_bytes(65) == 0
    assert lib.qd_episode_stats_workspace_bytes(4096) == 64 * 4096 * 2 * (8 + 4) + 65 * 16 * 8 * 8 and lib.qd_episode_stats_workspace_bytes(0) == 0
    dummy = C.c_void_p(256)
    assert lib.qd_column_stats(dummy, 10, 65, dummy, dummy, 1 << 30, None) == L.QD_ERR_UNSUPPORTED and "64 columns" in L.last_error()
    assert lib.qd_column_stats(dummy, 0, 22, dummy, dummy, 1 << 30, None) == L.QD_ERR_INVALID and "empty batch" in L.last_error()
    assert lib.qd_column_stats(dummy, 10, 22, dummy, dummy, 8, None) == L.QD_ERR_ARENA
    assert lib.qd_episode_stats(dummy, dummy, 4, 8, None, dummy, dummy, 1 << 20, None) == L.QD_ERR_INVALID


def test_merge_of_shard_statistics_equals_statistics_of_the_whole(SG):
    """config 4: every rank computes the statistics of its own fragment; the learner merges them (host arithmetic only, so this
    runs without a GPU: the shard statistics here come from the oracle)"""
    from oracle import stats_ref as S
    import importlib
    x = SG["batch_obs"].astype(np.float64)
    cuts = [0, 700, 701, 2100, len(x)]
    parts = [S.column_stats(x[a:b]) for a, b in zip(cuts, cuts[1:])]
    rows = [b - a for a, b in zip(cuts, cuts[1:])]
    try:
        merge = importlib.import_module("mujoco_drone_amd.custom_logging").merge_column_stats
    except ImportError as ex:                                  # the module needs the built library; the merge itself does not
        pytest.skip(str(ex))
    got, want = merge(parts, rows), S.column_stats(x)
    for k in ("min", "max"):
        np.testing.assert_array_equal(got[k], want[k])
    np.testing.assert_allclose(got["mean"], want["mean"], rtol=1e-13)
    np.testing.assert_allclose(got["var"], want["var"], rtol=1e-11)
    with pytest.raises(ValueError):
        merge(parts, rows[:-1])
