"""CPU tier: the C restatement (oracle/pair_oracle.c) against the golden kernel KATs."""
import numpy as np
import pytest

from oracle import c_oracle


@pytest.mark.skipif(not c_oracle.available(), reason="oracle/libpair_oracle.so not built")
def test_c_oracle_matches_goldens(g1_cases):
    assert c_oracle.threads() >= 1
    for name, c in g1_cases.items():
        vc = float(c["v_core"]) if bool(c["viscous"]) else 0.0
        if name == "p3x3_inviscid_self":
            continue
        u, w = c_oracle.induced_velocity(c["g"].astype(float), c["xw"], c["zw"], c["xp"], c["zp"], vc)
        # same terms; only the order of the row sum differs (sequential vs NumPy pairwise)
        scale = np.abs(c["g"]).sum() if c["g"].size > 1 else 1.0
        bound = 1e-13 * max(1.0, np.abs(c["u"]).max(), np.abs(c["w"]).max()) * max(1.0, scale)
        np.testing.assert_allclose(u, c["u"], rtol=1e-11, atol=bound, err_msg=name)
        np.testing.assert_allclose(w, c["w"], rtol=1e-11, atol=bound, err_msg=name)


@pytest.mark.skipif(not c_oracle.available(), reason="oracle/libpair_oracle.so not built")
def test_c_oracle_inviscid_self_pair_nan(g1_cases):
    c = g1_cases["p3x3_inviscid_self"]
    u, w = c_oracle.induced_velocity(c["g"], c["xw"], c["zw"], c["xp"], c["zp"], 0.0)
    assert np.array_equal(np.isnan(u), np.isnan(c["u"])) and np.array_equal(np.isnan(w), np.isnan(c["w"]))
