"""CPU: the C oracle (oracle/lcp_oracle.c) against the golden vectors produced by the reference."""
import numpy as np
import pytest

from helpers import lcp_goldens, load_lcp, rel
from oracle import lcp_oracle as O


@pytest.mark.parametrize("path", lcp_goldens(), ids=lambda p: p.split("/")[-1][:-4])
def test_oracle_forward_matches_reference(path):
    g = load_lcp(path)
    z, lam, s, nu, it, st = O.forward(g["Q"], g["p"], g["G"], g["h"], g["A"], g["b"], g["F"], max_iter=int(g["max_iter"]))
    assert (st == (4 if "inaccurate" in g else 0)).all()      # DSS_LCP_INACCURATE: the INACC_ERR condition (batch.py:165-167)
    # primal solution (velocities): unique -> tight.  Multipliers are only unique up to the null space
    # of G^T on redundant contact manifolds; the reference itself moves by ~1e-5 there under 1-ulp changes.
    assert rel(z, g["zhat"]) < 1e-10
    assert rel(s, g["slack"]) < 1e-6
    assert rel(lam, g["lam"]) < 1e-3


@pytest.mark.parametrize("path", lcp_goldens(), ids=lambda p: p.split("/")[-1][:-4])
def test_oracle_backward_matches_reference(path):
    g = load_lcp(path)
    # backward as a pure function of the reference's own forward state
    out = O.backward(g["Q"], g["G"], g["A"], g["F"], g["zhat"], g["lam"], g["slack"], g["nu"], g["dl_dz"])
    for name, got in zip("QpGhAbF", out):
        want = g["d" + name]
        if want.size:
            assert rel(got, want) < 1e-7, name


def test_oracle_flags_non_spd():
    Q = -np.eye(4)[None]
    z, lam, s, nu, it, st = O.forward(Q, np.zeros((1, 4)), np.ones((1, 2, 4)), np.ones((1, 2)), np.zeros((1, 0, 4)),
                                      np.zeros((1, 0)), np.zeros((1, 2, 2)))
    assert st[0] == 1


def test_oracle_on_config1_calls():
    """BASELINE configs[0] (2-D Circle on Rect, the reference's own CPU case): every LCP call of the rollout."""
    from helpers import config1_calls
    for c in config1_calls():
        z, lam, s, nu, it, st = O.forward(c["Q"], c["p"], c["G"], c["h"], c["A"], c["b"], c["F"], max_iter=c["max_iter"])
        assert rel(z, c["z"]) < 1e-10 and rel(lam, c["lam"]) < 1e-6
