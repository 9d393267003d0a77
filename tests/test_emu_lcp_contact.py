"""CPU: logic of csrc/lcp_contact.hip (fiber emulator) vs the dense C oracle on expanded operands."""
import numpy as np
import pytest

import structured as S
from emu import emu
from helpers import rel
from oracle import lcp_oracle as O


@pytest.mark.parametrize("cfg", [dict(seed=1, B=3, nb=2, maxc=8, fd=8), dict(seed=2, B=2, nb=4, maxc=16, fd=8),
                                 dict(seed=3, B=2, nb=3, maxc=8, fd=4), dict(seed=4, B=1, nb=3, maxc=80, fd=8, nc_lo=70),
                                 dict(seed=6, B=2, nb=8, maxc=24, fd=8, nc_lo=10),
                                 dict(seed=7, B=1, nb=3, maxc=136, fd=8, nc_lo=100),   # > 128: the streaming kernel
                                 dict(seed=8, B=1, nb=8, maxc=16, fd=8, nc_lo=8, rot_A=True)])   # equality rows that are not the identity: the unreduced KKT system
def test_contact_lcp_forward_backward_vs_dense_oracle(cfg):
    cfg = dict(cfg)
    rot_A = cfg.pop("rot_A", False)
    P = S.random_problem(**cfg)
    if rot_A:   # body 0 still pinned, but by a rotated set of rows: the pinned-body shortcut must not trigger
        Qr, _ = np.linalg.qr(np.random.default_rng(77).standard_normal((6, 6)))
        P["A"][:, :, :6] = Qr
    x, lam, slack, nu, it, st = emu.lcp_contact_forward(P, max_iter=10)
    dl = np.random.default_rng(9).standard_normal(x.shape)
    dM, dp, dcop, dA, db = emu.lcp_contact_backward(P, x, lam, slack, nu, dl)
    for s in range(P["Mblk"].shape[0]):
        nc, fd = int(P["nc"][s]), P["fd"]
        Q, p, G, h, A, b, F = S.expand_dense(P, s)
        zo, lo, so, nuo, ito, sto = O.forward(Q[None], p[None], G[None], h[None], A[None], b[None], F[None], max_iter=10)
        assert abs(int(ito[0]) - int(it[s])) <= 1, (ito, it)
        assert rel(x[s], zo[0]) < 1e-9
        assert rel(S.struct_vec(slack[s], nc, fd), so[0]) < 1e-6
        assert rel(S.struct_vec(lam[s], nc, fd), lo[0]) < 1e-5
        assert rel(nu[s], nuo[0]) < 1e-8
        # backward as a pure function of the same forward state
        ls, ss = S.struct_vec(lam[s], nc, fd), S.struct_vec(slack[s], nc, fd)
        dQ, dpo, dG, dh, dAo, dbo, dF = O.backward(Q[None], G[None], A[None], F[None], x[s][None], ls[None], ss[None], nu[s][None], dl[s][None])
        wM, wp, wcop = S.contract_dense_grads(P, s, dQ[0], dpo[0], dG[0], dh[0], dF[0])
        assert rel(dM[s], wM) < 1e-6
        assert rel(dp[s], wp) < 1e-6
        assert rel(dcop[s], wcop) < 1e-6
        assert rel(dA[s], dAo[0]) < 1e-6 and rel(db[s], dbo[0]) < 1e-6


def test_no_contacts_is_plain_linear_solve():
    P = S.random_problem(seed=5, B=2, nb=2, maxc=4, fd=8)
    P["nc"][:] = 0
    x, lam, slack, nu, it, st = emu.lcp_contact_forward(P)
    for s in range(2):
        Q, p, G, h, A, b, F = S.expand_dense(P, s)
        K = np.block([[Q, A.T], [A, np.zeros((6, 6))]])
        sol = np.linalg.solve(K, np.concatenate([-p, b]))
        assert rel(x[s], sol[:12]) < 1e-12 and it[s] == 0


def test_block_tridiagonal_elimination_and_its_fallback_under_the_emulator():
    """(tests/test_lcp_contact_gpu.py has the device version.)  Chain-structured contacts -> regk_factor_lead_tri; one contact that
    skips a body -> the general elimination; both against the dense oracle."""
    P = S.random_problem(seed=33, B=2, nb=8, maxc=24, fd=8, nc_lo=10, chain=True)
    P["cbody"][1, :, 0] = (1, 3)
    x, lam, slack, nu, it, st = emu.lcp_contact_forward(P, max_iter=10)
    for s in range(2):
        Q, p, G, h, A, b, F = S.expand_dense(P, s)
        zo = O.forward(Q[None], p[None], G[None], h[None], A[None], b[None], F[None], max_iter=10)[0]
        assert rel(x[s], zo[0]) < 1e-9, (s, rel(x[s], zo[0]))
