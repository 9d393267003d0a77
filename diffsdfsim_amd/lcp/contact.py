"""Raw device calls of the contact-structured LCP (csrc/lcp_contact.hip, C ABI section B2).

Operand layout is documented in include/diffsdfsim_hip.h.  All tensors live on the HIP device;
there is no CPU path.
"""
import ctypes

import torch

from .. import _lib


def _setup(L):
    L.dss_lcp_contact_workspace_bytes.restype = ctypes.c_size_t


def lcp_contact_forward(Mblk, pvec, A, bvec, cop, cbody, nc, fric_dirs, eps=1e-12, not_improved_lim=3, max_iter=10,
                        workspace=None, active=None):
    """-> x [B,nz], lam [B,NR,maxc], slack [B,NR,maxc], nu [B,neq], iters [B], status [B]."""
    _lib.require_device(Mblk, pvec, cop, cbody, nc)
    L = _lib.lib()
    _setup(L)
    B, nb = Mblk.shape[0], Mblk.shape[1]
    neq = A.shape[1] if A is not None and A.numel() else 0
    maxc = cop.shape[2]
    NR = fric_dirs + 2
    dev = Mblk.device
    f64 = dict(dtype=torch.float64, device=dev)
    x = torch.empty(B, 6 * nb, **f64)
    lam = torch.zeros(B, NR, maxc, **f64)
    slack = torch.zeros(B, NR, maxc, **f64)
    nu = torch.zeros(B, neq, **f64)
    iters = torch.zeros(B, dtype=torch.int32, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    nbytes = L.dss_lcp_contact_workspace_bytes(B, nb, neq, maxc, fric_dirs)
    if workspace is None or workspace.numel() < nbytes:
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    rc = L.dss_lcp_contact_forward(_lib.ptr(Mblk), _lib.ptr(pvec), _lib.ptr(A), _lib.ptr(bvec), _lib.ptr(cop),
                                   _lib.ptr(cbody), _lib.ptr(nc), _lib.ptr(active), B, nb, neq, maxc, fric_dirs, ctypes.c_double(eps),
                                   int(not_improved_lim), int(max_iter), _lib.ptr(x), _lib.ptr(lam), _lib.ptr(slack),
                                   _lib.ptr(nu), _lib.ptr(iters), _lib.ptr(status), _lib.ptr(workspace),
                                   ctypes.c_size_t(workspace.numel()), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_lcp_contact_forward")
    return x, lam, slack, nu, iters, status


def lcp_contact_backward(Mblk, A, cop, cbody, nc, fric_dirs, x, lam, slack, nu, dl_dx, want_dA=False):
    """-> dMblk [B,nb,6,6], dpvec [B,nz], dcop like cop, (dA, db | None, None)."""
    _lib.require_device(Mblk, cop, x, dl_dx)
    L = _lib.lib()
    B, nb = Mblk.shape[0], Mblk.shape[1]
    neq = A.shape[1] if A is not None and A.numel() else 0
    maxc = cop.shape[2]
    dev = Mblk.device
    f64 = dict(dtype=torch.float64, device=dev)
    dM = torch.empty(B, nb, 6, 6, **f64)
    dp = torch.empty(B, 6 * nb, **f64)
    dcop = torch.empty_like(cop)
    dA = torch.empty(B, neq, 6 * nb, **f64) if want_dA and neq else None
    db = torch.empty(B, neq, **f64) if want_dA and neq else None
    rc = L.dss_lcp_contact_backward(_lib.ptr(Mblk), _lib.ptr(A), _lib.ptr(cop), _lib.ptr(cbody), _lib.ptr(nc), None, B, nb, neq,
                                    maxc, fric_dirs, _lib.ptr(x), _lib.ptr(lam), _lib.ptr(slack), _lib.ptr(nu),
                                    _lib.ptr(dl_dx), _lib.ptr(dM), _lib.ptr(dp), _lib.ptr(dcop), _lib.ptr(dA), _lib.ptr(db),
                                    _lib.stream_ptr(dev))
    _lib.check(rc, "dss_lcp_contact_backward")
    return dM, dp, dcop, dA, db
