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


@pytest.mark.parametrize("tag", ["rma_full", "rma_model", "simple_mlp"])
def test_policy_oracle_vs_reference_models(PG, tag):
    from oracle import policy_ref as P
    w = weights_of(PG, tag)
    logits, value = P.FAMILIES[tag](w, PG["obs"], PG["prev_actions"])
    np.testing.assert_allclose(logits, PG[tag + "_logits"], atol=3e-6)       # the reference runs in float32
    np.testing.assert_allclose(value, PG[tag + "_value"], atol=3e-6)
    np.testing.assert_allclose(P.beta_mean_action(PG[tag + "_logits"]), PG[tag + "_action"], atol=1e-6)
    np.testing.assert_allclose(P.beta_logp(PG[tag + "_logits"], PG[tag + "_action"]), PG[tag + "_logp"], atol=2e-5)
