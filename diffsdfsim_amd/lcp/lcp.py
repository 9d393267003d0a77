"""``LCPFunction`` -- drop-in for ``lcp_physics.lcp.lcp.LCPFunction`` (reference lcp.py:43-214).

Same factory signature, same call signature ``fn(Q, p, G, h, A, b, F) -> zhat [nBatch, nz]``,
same broadcasting of un-batched operands (util.py:70-83, their gradients are batch means,
lcp.py:187-208), same exceptions.  Forward and backward run in the HIP library
(csrc/lcp_dense.hip) through the C ABI; there is no CPU path.

Differences that are by design:
  * re-entrant: no module globals, nothing is mutated in backward (the reference caches a
    batched eye in a global and overwrites the saved S_LU, batch.py:482-494, lcp.py:178);
  * the backward re-factors instead of keeping Q_LU / S_LU / R alive on the tape;
  * batch-wide termination tests are applied per system.
"""
import ctypes

import torch
from torch.autograd import Function

from .. import _lib

INACC_ERR = """
--------
qpth warning: Returning an inaccurate and potentially incorrect solution.

Some residual is large.
Your problem may be infeasible or difficult.
--------
"""

Q_LU_ERR = """
qpth Error: Cannot perform LU factorization on Q.
Please make sure that your Q matrix is PSD and has
a non-zero diagonal.
"""

_DIMS = (3, 2, 3, 2, 3, 2, 3)


def _n_batch(params):
    for prm, dim in zip(params, _DIMS):
        if prm.ndimension() == dim:
            return prm.size(0)
    return 1


def _expand(X, nBatch, nDim):
    if X.ndimension() in (0, nDim) or X.nelement() == 0:
        return X, False
    if X.ndimension() == nDim - 1:
        return X.unsqueeze(0).expand(*([nBatch] + list(X.size()))), True
    raise RuntimeError("Unexpected number of dimensions.")


def _dev(t):
    return t.detach().contiguous().to(torch.float64)


def lcp_dense_forward(Q, p, G, h, A, b, F, eps, not_improved_lim, max_iter, check_spd):
    """Raw batched call into the C ABI.  All operands [B, ...] float64 on the HIP device."""
    _lib.require_device(Q, p, G, h, F)
    L = _lib.lib()
    B, nineq, nz = G.shape
    neq = A.shape[1] if A.nelement() > 0 else 0
    dev = Q.device
    zhat = torch.empty(B, nz, dtype=torch.float64, device=dev)
    lam = torch.empty(B, nineq, dtype=torch.float64, device=dev)
    slack = torch.empty(B, nineq, dtype=torch.float64, device=dev)
    nu = torch.empty(B, neq, dtype=torch.float64, device=dev)
    iters = torch.zeros(B, dtype=torch.int32, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    nbytes = L.dss_lcp_dense_workspace_bytes(B, nz, nineq, neq)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    rc = L.dss_lcp_dense_forward(_lib.ptr(Q), _lib.ptr(p), _lib.ptr(G), _lib.ptr(h), _lib.ptr(A), _lib.ptr(b),
                                 _lib.ptr(F), B, nz, nineq, neq, ctypes.c_double(eps), int(not_improved_lim),
                                 int(max_iter), int(bool(check_spd)), _lib.ptr(zhat), _lib.ptr(lam), _lib.ptr(slack),
                                 _lib.ptr(nu), _lib.ptr(iters), _lib.ptr(status), _lib.ptr(ws),
                                 ctypes.c_size_t(nbytes), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_lcp_dense_forward")
    return zhat, lam, slack, nu, iters, status


def lcp_dense_backward(Q, G, A, F, zhat, lam, slack, nu, dl_dz):
    _lib.require_device(Q, G, F, zhat, dl_dz)
    L = _lib.lib()
    B, nineq, nz = G.shape
    neq = A.shape[1] if A.nelement() > 0 else 0
    dev = Q.device
    mk = lambda *s: torch.empty(*s, dtype=torch.float64, device=dev)
    dQ, dp, dG, dh = mk(B, nz, nz), mk(B, nz), mk(B, nineq, nz), mk(B, nineq)
    dA, db, dF = mk(B, neq, nz), mk(B, neq), mk(B, nineq, nineq)
    nbytes = L.dss_lcp_dense_workspace_bytes(B, nz, nineq, neq)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    rc = L.dss_lcp_dense_backward(_lib.ptr(Q), _lib.ptr(G), _lib.ptr(A), _lib.ptr(F), B, nz, nineq, neq,
                                  _lib.ptr(zhat), _lib.ptr(lam), _lib.ptr(slack), _lib.ptr(nu), _lib.ptr(dl_dz),
                                  _lib.ptr(dQ), _lib.ptr(dp), _lib.ptr(dG), _lib.ptr(dh), _lib.ptr(dA), _lib.ptr(db),
                                  _lib.ptr(dF), _lib.ptr(ws), ctypes.c_size_t(nbytes), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_lcp_dense_backward")
    return dQ, dp, dG, dh, dA, db, dF


def LCPFunction(eps=1e-12, verbose=0, notImprovedLim=3, max_iter=20, solver=1, check_Q_spd=True):
    """Factory with the reference's signature (lcp.py:43-45).  ``solver`` must be 1 (PDIPM)."""
    if solver != 1:
        raise NotImplementedError("only the batched PDIPM solver (solver=1) exists on the HIP path; "
                                  "the reference's CVXPY alternative (solver=2) is out of scope")

    class LCPFunctionFn(Function):
        @staticmethod
        def forward(ctx, Q_, p_, G_, h_, A_, b_, F_):
            nBatch = _n_batch((Q_, p_, G_, h_, A_, b_, F_))
            Q, _ = _expand(Q_, nBatch, 3)
            p, _ = _expand(p_, nBatch, 2)
            G, _ = _expand(G_, nBatch, 3)
            h, _ = _expand(h_, nBatch, 2)
            A, _ = _expand(A_, nBatch, 3)
            b, _ = _expand(b_, nBatch, 2)
            F, _ = _expand(F_, nBatch, 3)
            _, nineq, nz = G.size()
            neq = A.size(1) if A.nelement() > 0 else 0
            assert neq > 0 or nineq > 0
            ctx.neq, ctx.nineq, ctx.nz = neq, nineq, nz
            Qc, pc, Gc, hc, Fc = _dev(Q), _dev(p), _dev(G), _dev(h), _dev(F)
            Ac, bc = (_dev(A), _dev(b)) if neq > 0 else (Qc.new_empty(0), Qc.new_empty(0))
            zhat, lam, slack, nu, iters, status = lcp_dense_forward(
                Qc, pc, Gc, hc, Ac, bc, Fc, eps, notImprovedLim, max_iter, check_Q_spd)
            st = int(status.max().item())  # one host sync; the reference syncs per sample (lcp.py:109-113)
            if st == 1:
                raise RuntimeError('Q is not SPD.')
            if st == 2:
                raise RuntimeError(Q_LU_ERR)
            if st == 4 and verbose >= 0:
                print(INACC_ERR)
            ctx.lams, ctx.nus, ctx.slacks, ctx.iters = lam, nu, slack, iters
            ctx.save_for_backward(zhat, Q_, p_, G_, h_, A_, b_, F_)
            return zhat.to(Q_.dtype)

        @staticmethod
        def backward(ctx, dl_dzhat):
            zhat, Q, p, G, h, A, b, F = ctx.saved_tensors
            nBatch = _n_batch((Q, p, G, h, A, b, F))
            Q, Q_e = _expand(Q, nBatch, 3)
            p, p_e = _expand(p, nBatch, 2)
            G, G_e = _expand(G, nBatch, 3)
            h, h_e = _expand(h, nBatch, 2)
            A, A_e = _expand(A, nBatch, 3)
            b, b_e = _expand(b, nBatch, 2)
            F, F_e = _expand(F, nBatch, 3)
            neq = ctx.neq
            Qc, Gc, Fc = _dev(Q), _dev(G), _dev(F)
            Ac = _dev(A) if neq > 0 else Qc.new_empty(0)
            dQ, dp, dG, dh, dA, db, dF = lcp_dense_backward(
                Qc, Gc, Ac, Fc, zhat, ctx.lams, ctx.slacks, ctx.nus, _dev(dl_dzhat))
            if F_e:
                dF = dF.mean(0)
            if G_e:
                dG = dG.mean(0)
            if h_e:
                dh = dh.mean(0)
            if neq > 0:
                if A_e:
                    dA = dA.mean(0)
                if b_e:
                    db = db.mean(0)
            else:
                dA, db = None, None
            if Q_e:
                dQ = dQ.mean(0)
            if p_e:
                dp = dp.mean(0)
            return dQ, dp, dG, dh, dA, db, dF

    return LCPFunctionFn.apply
