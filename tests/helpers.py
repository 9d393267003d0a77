import glob
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def lcp_goldens():
    return sorted(glob.glob(os.path.join(GOLDEN, "lcp_*.npz")))


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if b.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def load_lcp(path):
    g = dict(np.load(path))
    nB, nz = g["Q"].shape[0], g["Q"].shape[1]
    if g["A"].size == 0:
        g["A"] = np.zeros((nB, 0, nz))
        g["b"] = np.zeros((nB, 0))
        g["nu"] = np.zeros((nB, 0))
    if "dF" not in g:  # rank-one, not stored for the large cases (oracle/gen/gen_lcp_golden.py)
        g["dF"] = -g["dh"][:, :, None] * g["lam"][:, None, :]
    return g


def random_lcp(seed, nB, nz, nineq, neq, with_F=True):
    """Seeded random dense LCP with SPD Q (float64 numpy)."""
    r = np.random.default_rng(seed)
    L = r.standard_normal((nB, nz, nz))
    Q = L @ L.transpose(0, 2, 1) + np.eye(nz)
    p = r.standard_normal((nB, nz))
    G = r.standard_normal((nB, nineq, nz))
    h = r.random((nB, nineq))
    A = r.standard_normal((nB, neq, nz))
    b = r.standard_normal((nB, neq)) * 0.1
    Fh = r.standard_normal((nB, nineq, nineq)) * 0.1
    F = Fh @ Fh.transpose(0, 2, 1) + 0.05 * r.standard_normal((nB, nineq, nineq)) if with_F else np.zeros((nB, nineq, nineq))
    return Q, p, G, h, A, b, F


def config1_calls():
    """LCP calls recorded from the reference's 2-D config-1 rollout (oracle/gen/gen_config1_golden.py)."""
    g = np.load(os.path.join(GOLDEN, "config1_lcp.npz"))
    out = []
    for i in range(int(g["n_calls"])):
        c = {k: g["c%d_%s" % (i, k)] for k in ("Q", "p", "G", "h", "A", "b", "F", "z", "lam", "slack", "nu", "dl")}
        c["max_iter"] = int(g["c%d_max_iter" % i])
        out.append(c)
    return out
