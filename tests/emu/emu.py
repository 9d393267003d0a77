"""ctypes front end of the CPU emulation build of the kernels (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    subprocess.check_call(["make", "-s", "-C", _HERE])
    if _LIB is None:
        _LIB = ctypes.CDLL(os.path.join(_HERE, "_build", "libdss_emu.so"))
        _LIB.dss_lcp_dense_workspace_bytes.restype = ctypes.c_size_t
    return _LIB


import numpy as np  # noqa: E402


def _p(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None and a.size else ctypes.c_void_p(0)


def _c(a, dtype=np.float64):
    return np.ascontiguousarray(np.asarray(a, dtype=dtype))


def lcp_dense_forward(Q, p, G, h, A, b, F, eps=1e-12, nil=3, max_iter=20, check_spd=True):
    L = lib()
    Q, p, G, h, A, b, F = (_c(x) for x in (Q, p, G, h, A, b, F))
    B, nineq, nz = G.shape
    neq = A.shape[1] if A.size else 0
    zhat = np.zeros((B, nz)); lam = np.zeros((B, nineq)); slack = np.zeros((B, nineq)); nu = np.zeros((B, neq))
    iters = np.zeros(B, np.int32); status = np.zeros(B, np.int32)
    nbytes = L.dss_lcp_dense_workspace_bytes(B, nz, nineq, neq)
    ws = np.zeros(nbytes, np.uint8)
    rc = L.dss_lcp_dense_forward(_p(Q), _p(p), _p(G), _p(h), _p(A), _p(b), _p(F), B, nz, nineq, neq,
                                 ctypes.c_double(eps), nil, max_iter, int(check_spd), _p(zhat), _p(lam), _p(slack),
                                 _p(nu), _p(iters), _p(status), _p(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0, rc
    return zhat, lam, slack, nu, iters, status


def lcp_dense_backward(Q, G, A, F, zhat, lam, slack, nu, dl):
    L = lib()
    Q, G, A, F, zhat, lam, slack, nu, dl = (_c(x) for x in (Q, G, A, F, zhat, lam, slack, nu, dl))
    B, nineq, nz = G.shape
    neq = A.shape[1] if A.size else 0
    mk = lambda *s: np.zeros(s)
    dQ, dp, dG, dh, dA, db, dF = mk(B, nz, nz), mk(B, nz), mk(B, nineq, nz), mk(B, nineq), mk(B, neq, nz), mk(B, neq), mk(B, nineq, nineq)
    nbytes = L.dss_lcp_dense_workspace_bytes(B, nz, nineq, neq)
    ws = np.zeros(nbytes, np.uint8)
    rc = L.dss_lcp_dense_backward(_p(Q), _p(G), _p(A), _p(F), B, nz, nineq, neq, _p(zhat), _p(lam), _p(slack), _p(nu),
                                  _p(dl), _p(dQ), _p(dp), _p(dG), _p(dh), _p(dA), _p(db), _p(dF), _p(ws),
                                  ctypes.c_size_t(nbytes), None)
    assert rc == 0, rc
    return dQ, dp, dG, dh, dA, db, dF


def lcp_contact_forward(P, eps=1e-12, nil=3, max_iter=10):
    L = lib()
    L.dss_lcp_contact_workspace_bytes.restype = ctypes.c_size_t
    B, nb, neq, maxc, fd = P["Mblk"].shape[0], P["nb"], P["neq"], P["maxc"], P["fd"]
    NR = fd + 2; nz = 6 * nb
    Mblk, pvec, A, bvec, cop = (_c(P[k]) for k in ("Mblk", "pvec", "A", "bvec", "cop"))
    cbody, nc = _c(P["cbody"], np.int32), _c(P["nc"], np.int32)
    x = np.zeros((B, nz)); lam = np.zeros((B, NR, maxc)); slack = np.zeros((B, NR, maxc)); nu = np.zeros((B, neq))
    iters = np.zeros(B, np.int32); status = np.zeros(B, np.int32)
    nbytes = L.dss_lcp_contact_workspace_bytes(B, nb, neq, maxc, fd)
    ws = np.zeros(nbytes, np.uint8)
    rc = L.dss_lcp_contact_forward(_p(Mblk), _p(pvec), _p(A), _p(bvec), _p(cop), _p(cbody), _p(nc), None, B, nb, neq, maxc, fd,
                                   ctypes.c_double(eps), nil, max_iter, _p(x), _p(lam), _p(slack), _p(nu), _p(iters),
                                   _p(status), _p(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0, rc
    return x, lam, slack, nu, iters, status


def lcp_contact_backward(P, x, lam, slack, nu, dl):
    L = lib()
    B, nb, neq, maxc, fd = P["Mblk"].shape[0], P["nb"], P["neq"], P["maxc"], P["fd"]
    nz = 6 * nb; NF = 3 * (1 + fd // 2) + 8
    Mblk, A, cop = (_c(P[k]) for k in ("Mblk", "A", "cop"))
    cbody, nc = _c(P["cbody"], np.int32), _c(P["nc"], np.int32)
    x, lam, slack, nu, dl = (_c(v) for v in (x, lam, slack, nu, dl))
    dM = np.zeros((B, nb, 6, 6)); dp = np.zeros((B, nz)); dcop = np.zeros((B, NF, maxc))
    dA = np.zeros((B, neq, nz)); db = np.zeros((B, neq))
    rc = L.dss_lcp_contact_backward(_p(Mblk), _p(A), _p(cop), _p(cbody), _p(nc), None, B, nb, neq, maxc, fd, _p(x), _p(lam),
                                    _p(slack), _p(nu), _p(dl), _p(dM), _p(dp), _p(dcop), _p(dA), _p(db), None)
    assert rc == 0, rc
    return dM, dp, dcop, dA, db


class EmuBackend:
    """numpy arrays + the CPU emulation build; plugs into diffsdfsim_amd.engine.BatchEngine for logic tests."""

    def __init__(self):
        self.lib = lib()

    def zeros(self, shape, dtype):
        return np.zeros(shape, dtype)

    def from_numpy(self, a):
        return np.ascontiguousarray(a).copy()

    def to_numpy(self, t):
        return t

    def ptr(self, t):
        return t.ctypes.data

    def stream(self):
        return None

    def read_int(self, t):
        return int(t.reshape(-1)[0])


def igr_query(pts, latent, Ws, bs, wrt="xyz"):
    L = lib()
    H = 128
    packed = np.zeros((7, 8, 32, 64)); bh = np.zeros((7, H)); lane = np.arange(64)
    for l in range(1, 8):
        W = np.zeros((H, H)); W[: Ws[l].shape[0]] = Ws[l]; bh[l - 1, : len(bs[l])] = bs[l]
        for t in range(8):
            for ks in range(32):
                packed[l - 1, t, ks] = W[16 * t + (lane & 15), 4 * ks + (lane >> 4)]
    pts = _c(pts); n = len(pts)
    sdf = np.zeros(n); grad = np.zeros((n, 3))
    W0, b0, W8, b8, lat = _c(Ws[0]), _c(bs[0]), _c(Ws[8][0]), _c(bs[8]), _c(latent)
    fn = L.dss_igr_query_latent_grad if wrt == "latent" else L.dss_igr_query
    rc = fn(_p(pts), _p(lat), _p(W0), _p(b0), _p(packed), _p(bh), _p(W8), _p(b8), n, _p(sdf), _p(grad), None)
    assert rc == 0
    return sdf, grad


def sdf_query(shape_type, prm, pts):
    L = lib()
    pts = _c(pts); n = len(pts)
    prm = _c(np.concatenate([np.asarray(prm, np.float64).reshape(-1), np.zeros(4)])[:4])
    sdf = np.zeros(n); grad = np.zeros((n, 3)); mask = np.zeros(n, np.uint8)
    rc = L.dss_sdf_query(int(shape_type), _p(prm), _p(pts), n, _p(sdf), _p(grad), _p(mask), None)
    assert rc == 0
    return sdf, grad, mask.astype(bool)


def grid_sdf_query(grid, scale, pts):
    L = lib()
    G = _c(grid); pts = _c(pts); n = len(pts)
    sdf = np.zeros(n); grad = np.zeros((n, 3)); mask = np.zeros(n, np.uint8)
    rc = L.dss_grid_sdf_query(_p(G), G.shape[0], G.shape[1], G.shape[2], ctypes.c_double(scale), _p(pts), n, _p(sdf), _p(grad),
                              _p(mask), None)
    assert rc == 0
    return sdf, grad, mask.astype(bool)


def mesh_inertia(verts, faces, mass):
    L = lib()
    V = _c(verts); F = _c(faces, np.int32)
    voff = np.zeros(1, np.int32); foff = np.zeros(1, np.int32); nf = np.array([len(F)], np.int32)
    M = np.array([mass], np.float64); J = np.zeros(9); vol = np.zeros(1)
    rc = L.dss_mesh_inertia(_p(V), _p(F), _p(voff), _p(foff), _p(nf), 1, _p(M), _p(J), _p(vol), None)
    assert rc == 0
    return J.reshape(3, 3), vol[0]


def marching_cubes(phi, iso=0.0):
    import sys as _sys
    _sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    from diffsdfsim_amd import mc_tables
    L = lib()
    ntri, tri, _ = mc_tables.tables()
    phi = _c(phi); n0, n1, n2 = phi.shape
    L.dss_mc_workspace_bytes.restype = ctypes.c_size_t
    nb = L.dss_mc_workspace_bytes(n0, n1, n2)
    ws = np.zeros(nb, np.uint8); tot = np.zeros(2, np.int32)
    ntri_c = _c(ntri, np.int32); tri_c = _c(tri, np.int8)
    rc = L.dss_mc_count(_p(phi), n0, n1, n2, ctypes.c_double(iso), _p(ntri_c), _p(ws), ctypes.c_size_t(nb), _p(tot), None)
    assert rc == 0
    V = np.zeros((max(int(tot[0]), 1), 3)); F = np.zeros((max(int(tot[1]), 1), 3), np.int32)
    rc = L.dss_mc_emit(_p(phi), n0, n1, n2, ctypes.c_double(iso), _p(ntri_c), _p(tri_c), mc_tables.MAX_TRI, _p(ws), _p(V), _p(F), None)
    assert rc == 0
    return V[: tot[0]], F[: tot[1]]


def meshsdf_backward(shape_type, unit_prm, unit_verts, gbar):
    L = lib()
    k = np.asarray(unit_prm).size
    prm = _c(np.concatenate([np.asarray(unit_prm, np.float64).reshape(-1), np.zeros(4)])[:4])
    V = _c(unit_verts); Gb = _c(gbar); out = np.zeros(4)
    rc = L.dss_meshsdf_backward(int(shape_type), _p(prm), _p(V), _p(Gb), len(V), _p(out), None)
    assert rc == 0
    return out[:max(k, 3)]


def mesh_inertia_backward(verts, faces, mass, gJ):
    L = lib()
    V = _c(verts); F = _c(faces, np.int32); g = _c(gJ).reshape(9); out = np.zeros_like(V)
    rc = L.dss_mesh_inertia_backward(_p(V), _p(F), len(V), len(F), ctypes.c_double(mass), _p(g), _p(out), None)
    assert rc == 0
    return out


def contacts2d_forward(kind, nv, pos, rad, verts, sat_in, eps):
    """dss_contacts2d_forward on host arrays: kind / nv / sat_in [2][P], pos [2][P][2], rad [2][P], verts [2][P][maxv][2]."""
    L = lib()
    kind, nv, sat_in = (_c(x, np.int32) for x in (kind, nv, sat_in))
    pos, rad, verts = (_c(x) for x in (pos, rad, verts))
    P, maxv = pos.shape[1], verts.shape[2]
    sat_out = np.zeros((2, P), np.int32); count = np.zeros(P, np.int32); out = np.zeros((P, 2, 7))
    rc = L.dss_contacts2d_forward(P, maxv, _p(kind), _p(nv), _p(pos), _p(rad), _p(verts), _p(sat_in), ctypes.c_double(eps),
                                  _p(sat_out), _p(count), _p(out), None)
    assert rc == 0, rc
    return out, count, sat_out


def contacts2d_backward(kind, nv, pos, rad, verts, sat_in, eps, gout):
    L = lib()
    kind, nv, sat_in = (_c(x, np.int32) for x in (kind, nv, sat_in))
    pos, rad, verts, gout = (_c(x) for x in (pos, rad, verts, gout))
    P, maxv = pos.shape[1], verts.shape[2]
    g_pos, g_rad, g_verts = np.zeros_like(pos), np.zeros_like(rad), np.zeros_like(verts)
    rc = L.dss_contacts2d_backward(P, maxv, _p(kind), _p(nv), _p(pos), _p(rad), _p(verts), _p(sat_in), ctypes.c_double(eps),
                                   _p(gout), _p(g_pos), _p(g_rad), _p(g_verts), None)
    assert rc == 0, rc
    return g_pos, g_rad, g_verts
