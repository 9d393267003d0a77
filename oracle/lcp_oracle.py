"""ctypes front end of the C oracle (TEST INFRASTRUCTURE; see lcp_oracle.c for the citations).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liblcp_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    return a if shape is None else a.reshape(shape)


def forward(Q, p, G, h, A, b, F, eps=1e-12, not_improved_lim=3, max_iter=20, check_spd=True):
    """Batched LCP forward.  Returns zhat, lam, slack, nu, iters, status (numpy arrays)."""
    Q = _c(Q); nB, nz = Q.shape[0], Q.shape[1]
    G = _c(G); nineq = G.shape[1]
    A = _c(A); neq = A.shape[1] if A.size else 0
    p, h, F = _c(p), _c(h), _c(F)
    A = A if neq else np.zeros((nB, 0, nz)); b = _c(b) if neq else np.zeros((nB, 0))
    zhat = np.zeros((nB, nz)); lam = np.zeros((nB, nineq)); slack = np.zeros((nB, nineq)); nu = np.zeros((nB, neq))
    iters = np.zeros(nB, dtype=np.int32); status = np.zeros(nB, dtype=np.int32)
    lib().lcp_oracle_forward(_p(Q), _p(p), _p(G), _p(h), _p(A), _p(b), _p(F),
                             ctypes.c_int(nB), ctypes.c_int(nz), ctypes.c_int(nineq), ctypes.c_int(neq),
                             ctypes.c_double(eps), ctypes.c_int(not_improved_lim), ctypes.c_int(max_iter),
                             ctypes.c_int(int(check_spd)), _p(zhat), _p(lam), _p(slack), _p(nu), _p(iters), _p(status))
    return zhat, lam, slack, nu, iters, status


def backward(Q, G, A, F, zhat, lam, slack, nu, dl_dz):
    Q = _c(Q); nB, nz = Q.shape[0], Q.shape[1]
    G = _c(G); nineq = G.shape[1]
    A = _c(A); neq = A.shape[1] if A.size else 0
    A = A if neq else np.zeros((nB, 0, nz))
    F, zhat, lam, slack, dl_dz = _c(F), _c(zhat), _c(lam), _c(slack), _c(dl_dz)
    nu = _c(nu) if neq else np.zeros((nB, 0))
    dQ = np.zeros((nB, nz, nz)); dp = np.zeros((nB, nz)); dG = np.zeros((nB, nineq, nz)); dh = np.zeros((nB, nineq))
    dA = np.zeros((nB, neq, nz)); db = np.zeros((nB, neq)); dF = np.zeros((nB, nineq, nineq))
    lib().lcp_oracle_backward(_p(Q), _p(G), _p(A), _p(F), ctypes.c_int(nB), ctypes.c_int(nz), ctypes.c_int(nineq),
                              ctypes.c_int(neq), _p(zhat), _p(lam), _p(slack), _p(nu), _p(dl_dz),
                              _p(dQ), _p(dp), _p(dG), _p(dh), _p(dA), _p(db), _p(dF))
    return dQ, dp, dG, dh, dA, db, dF
