"""Contact-structured LCP operands <-> the reference dense (Q,p,G,h,A,b,F) (TEST INFRASTRUCTURE; used by tests and bench.py cpu_baseline)."""
import numpy as np


def friction_dirs(n, fd):
    """physics3d/world.py:84-94 and utils.py:247-256 (orthogonal)."""
    e = np.eye(3)[np.argmin(np.abs(n))]
    d1 = np.cross(e, n); d1 /= np.linalg.norm(d1)
    d2 = np.cross(d1, n); d2 /= np.linalg.norm(d2)
    dirs = [d1, d2]
    if fd == 8:
        d3 = d1 + d2; d3 /= np.linalg.norm(d3)
        d4 = np.cross(d3, n); d4 /= np.linalg.norm(d4)
        dirs += [d3, d4]
    return np.stack(dirs)


def random_problem(seed, B, nb, maxc, fd=8, nc_lo=1, nc_hi=None, fixed_body0=True, ragged=True, chain=False):
    """Random scenes: SPD 6x6 mass blocks, body 0 pinned by 6 equality rows, random contacts.  chain: every contact joins neighbours in
    the body order or a body and body 0 -- a stack: the shape lcp_contact.hip's block-tridiagonal elimination is selected for."""
    r = np.random.default_rng(seed)
    ND = fd // 2
    NF = 3 * (1 + ND) + 8
    nz = 6 * nb
    neq = 6 if fixed_body0 else 0
    nc_hi = maxc if nc_hi is None else nc_hi
    Mblk = np.zeros((B, nb, 6, 6))
    for s in range(B):
        for b in range(nb):
            L = r.standard_normal((3, 3))
            Mblk[s, b, :3, :3] = L @ L.T * 0.1 + 0.2 * np.eye(3)
            Mblk[s, b, 3:, 3:] = (0.5 + r.random()) * np.eye(3)
    pvec = r.standard_normal((B, nz))
    A = np.zeros((B, neq, nz)); bvec = np.zeros((B, neq))
    if neq:
        A[:, :, :6] = np.eye(6)
    cop = np.zeros((B, NF, maxc)); cbody = np.zeros((B, 2, maxc), np.int32); nc = np.zeros(B, np.int32)
    for s in range(B):
        k = int(r.integers(nc_lo, nc_hi + 1)) if ragged else nc_hi
        nc[s] = k
        for c in range(k):
            b1 = int(r.integers(0, nb)); b2 = int((b1 + 1 + r.integers(0, nb - 1)) % nb)
            if chain:
                if r.random() < 0.25:
                    b1, b2 = (0, int(r.integers(1, nb))) if r.random() < 0.5 else (int(r.integers(1, nb)), 0)
                else:
                    lo = int(r.integers(1, nb - 1)); b1, b2 = (lo, lo + 1) if r.random() < 0.5 else (lo + 1, lo)
            n = r.standard_normal(3); n /= np.linalg.norm(n)
            D = np.concatenate([n[None], friction_dirs(n, fd)])
            cop[s, :3 * (1 + ND), c] = D.reshape(-1)
            o = 3 * (1 + ND)
            cop[s, o:o + 3, c] = r.standard_normal(3) * 0.5
            cop[s, o + 3:o + 6, c] = r.standard_normal(3) * 0.5
            cop[s, o + 6, c] = 0.1 + 0.8 * r.random()
            cop[s, o + 7, c] = 0.2 * r.standard_normal()
            cbody[s, 0, c], cbody[s, 1, c] = b1, b2
    return dict(Mblk=Mblk, pvec=pvec, A=A, bvec=bvec, cop=cop, cbody=cbody, nc=nc, nb=nb, neq=neq, maxc=maxc, fd=fd)


def rows_of(cop_s, cbody_s, c, nb, fd):
    """The fd+2 dense G rows of contact c in structured row order [n, +D.., -D.., cone]."""
    ND = fd // 2
    o = 3 * (1 + ND)
    D = cop_s[:o, c].reshape(1 + ND, 3)
    p1, p2 = cop_s[o:o + 3, c], cop_s[o + 3:o + 6, c]
    b1, b2 = cbody_s[0, c], cbody_s[1, c]
    rows = np.zeros((fd + 2, 6 * nb))

    def row(d):
        r = np.zeros(6 * nb)
        r[6 * b1:6 * b1 + 3] += np.cross(p1, d); r[6 * b1 + 3:6 * b1 + 6] += d
        r[6 * b2:6 * b2 + 3] -= np.cross(p2, d); r[6 * b2 + 3:6 * b2 + 6] -= d
        return r
    rows[0] = row(D[0])
    for k in range(1, ND + 1):
        rows[k] = row(D[k]); rows[ND + k] = -rows[k]
    return rows


def dense_index(q, c, nc, fd):
    if q == 0:
        return c
    if q == fd + 1:
        return nc + nc * fd + c
    return nc + c * fd + (q - 1)


def expand_dense(P, s):
    """Scene s as the dense operands the reference engine would build (engines.py:57-79)."""
    nb, fd, neq = P["nb"], P["fd"], P["neq"]
    nc = int(P["nc"][s]); NR = fd + 2; nz = 6 * nb; nineq = nc * NR
    Q = np.zeros((nz, nz))
    for b in range(nb):
        Q[6 * b:6 * b + 6, 6 * b:6 * b + 6] = P["Mblk"][s, b]
    G = np.zeros((nineq, nz)); h = np.zeros(nineq); F = np.zeros((nineq, nineq))
    o = 3 * (1 + fd // 2)
    for c in range(nc):
        rows = rows_of(P["cop"][s], P["cbody"][s], c, nb, fd)
        for q in range(NR):
            G[dense_index(q, c, nc, fd)] = rows[q]
        h[c] = P["cop"][s, o + 7, c]
        g = dense_index(NR - 1, c, nc, fd)
        F[g, c] = P["cop"][s, o + 6, c]
        for q in range(1, NR - 1):
            F[dense_index(q, c, nc, fd), g] = 1.0
            F[g, dense_index(q, c, nc, fd)] = -1.0
    return Q, P["pvec"][s], G, h, P["A"][s], P["bvec"][s], F


def struct_vec(v_struct_s, nc, fd):
    """[NR][maxc] structured per-contact vector -> dense reference ordering (length nc*NR)."""
    NR = fd + 2
    out = np.zeros(nc * NR)
    for c in range(nc):
        for q in range(NR):
            out[dense_index(q, c, nc, fd)] = v_struct_s[q, c]
    return out


def dense_vec_to_struct(v, nc, fd, maxc):
    NR = fd + 2
    out = np.zeros((NR, maxc))
    for c in range(nc):
        for q in range(NR):
            out[q, c] = v[dense_index(q, c, nc, fd)]
    return out


def contract_dense_grads(P, s, dQ, dp, dG, dh, dF):
    """Contract the reference's dense gradients onto the structured operands (chain rule by hand)."""
    nb, fd = P["nb"], P["fd"]
    nc = int(P["nc"][s]); ND = fd // 2; NF = 3 * (1 + ND) + 8; o = 3 * (1 + ND)
    dM = np.stack([dQ[6 * b:6 * b + 6, 6 * b:6 * b + 6] for b in range(nb)])
    dcop = np.zeros((NF, P["maxc"]))
    cop, cb = P["cop"][s], P["cbody"][s]
    for c in range(nc):
        b1, b2 = cb[0, c], cb[1, c]
        D = cop[:o, c].reshape(1 + ND, 3); p1, p2 = cop[o:o + 3, c], cop[o + 3:o + 6, c]
        dD = np.zeros((1 + ND, 3)); dp1 = np.zeros(3); dp2 = np.zeros(3)
        for q in range(fd + 1):
            g = dG[dense_index(q, c, nc, fd)]
            k, sg = (0, 1.0) if q == 0 else ((q, 1.0) if q <= ND else (q - ND, -1.0))
            gw1, gu1, gw2, gu2 = g[6 * b1:6 * b1 + 3], g[6 * b1 + 3:6 * b1 + 6], g[6 * b2:6 * b2 + 3], g[6 * b2 + 3:6 * b2 + 6]
            dD[k] += sg * (np.cross(gw1, p1) + gu1 - np.cross(gw2, p2) - gu2)
            dp1 += sg * np.cross(D[k], gw1)
            dp2 -= sg * np.cross(D[k], gw2)
        dcop[:o, c] = dD.reshape(-1); dcop[o:o + 3, c] = dp1; dcop[o + 3:o + 6, c] = dp2
        dcop[o + 6, c] = dF[dense_index(fd + 1, c, nc, fd), c]
        dcop[o + 7, c] = dh[c]
    return dM, dp, dcop
