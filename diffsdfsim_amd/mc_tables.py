"""Marching-cubes case tables, generated instead of transcribed.

The reference meshes its SDFs with `ev_sdf_utils.marching_cubes` (`sdf_physics/physics3d/bodies.py:653-704`), an
un-vendored CUDA extension: its vertex / face order is implementation defined and cannot be pinned.
What the path needs is a closed, consistently oriented triangle mesh of the zero level set, so the 256-case table is
derived here from first principles:

  * on every cube face the cut edges are joined by segments; an ambiguous face (two diagonal corners inside) always
    separates its INSIDE corners -- a rule that depends on the face's own four signs only, so the two cubes sharing
    a face agree and the surface is watertight;
  * the segments of the six faces close into loops (every cut edge lies on exactly two faces); each loop is
    triangulated as a fan and oriented so that its normal points from inside (phi < iso) to outside.

Conventions: corner c = cx + 2 cy + 4 cz; edge e = 4 axis + idx with the edge running along `axis` from the corner
whose other two coordinates are (idx & 1, idx >> 1) in increasing axis order.  inside <=> phi < iso.
"""
import functools

import numpy as np

MAX_TRI = 8


def edge_corners(e):
    axis, idx = e // 4, e % 4
    others = [a for a in range(3) if a != axis]
    c0 = [0, 0, 0]
    c0[others[0]], c0[others[1]] = idx & 1, idx >> 1
    c1 = list(c0); c1[axis] = 1
    return tuple(c0), tuple(c1)


def _corner_id(c):
    return c[0] + 2 * c[1] + 4 * c[2]


def _edge_between(ca, cb):
    d = [abs(ca[i] - cb[i]) for i in range(3)]
    axis = d.index(1)
    lo = ca if ca[axis] == 0 else cb
    others = [a for a in range(3) if a != axis]
    return 4 * axis + lo[others[0]] + 2 * lo[others[1]]


def _faces():
    out = []
    for axis in range(3):
        o = [a for a in range(3) if a != axis]
        for side in (0, 1):
            cyc = []
            for u, v in ((0, 0), (1, 0), (1, 1), (0, 1)):
                c = [0, 0, 0]; c[axis] = side; c[o[0]] = u; c[o[1]] = v
                cyc.append(tuple(c))
            out.append(cyc)
    return out


def _share_face(e1, e2):
    pts = edge_corners(e1) + edge_corners(e2)
    return any(len({p[a] for p in pts}) == 1 for a in range(3))


def _triangulate(loop):
    """Triangulate the polygon `loop` (cube-edge ids) without a diagonal that joins two cut edges of one cube face:
    such a diagonal lies IN that face, where the neighbouring cube may put another one, and the surface would fold
    onto itself there.  Exhaustive search over the (few) triangulations."""
    def rec(poly):
        if len(poly) == 3:
            return [tuple(poly)]
        a, b = poly[0], poly[1]
        for k in range(2, len(poly)):      # triangle (a, b, poly[k]) splits the polygon
            c = poly[k]
            if k > 2 and _share_face(b, c):
                continue
            if k < len(poly) - 1 and _share_face(a, c):
                continue
            left = rec(poly[1:k + 1]) if k > 2 else []
            right = rec([poly[0]] + poly[k:]) if k < len(poly) - 1 else []
            if left is None or right is None:
                continue
            return [(a, b, c)] + left + right
        return None
    out = rec(list(loop))
    assert out is not None, loop
    return out


@functools.lru_cache(None)
def tables():
    """-> ntri [256] int32, tri [256, MAX_TRI, 3] int8 (edge ids, -1 padded), edge_mask [256] int32 (cut edges)."""
    ntri = np.zeros(256, np.int32); tri = -np.ones((256, MAX_TRI, 3), np.int8); emask = np.zeros(256, np.int32)
    for case in range(256):
        inside = lambda c: (case >> _corner_id(c)) & 1
        nxt = {}
        for axis in range(3):
            o = [a for a in range(3) if a != axis]
            for side in (0, 1):
                cyc = []
                for u, v in ((0, 0), (1, 0), (1, 1), (0, 1)):
                    c = [0, 0, 0]; c[axis] = side; c[o[0]] = u; c[o[1]] = v
                    cyc.append(tuple(c))
                # is `cyc` counter-clockwise when the face is seen from outside the cube?
                ccw = ((1 if axis != 1 else -1) * (1 if side else -1)) > 0
                flags = [inside(c) for c in cyc]
                e_of = lambda i: _edge_between(cyc[i % 4], cyc[(i + 1) % 4])     # edge between corner i and i+1
                cut = [i for i in range(4) if flags[i] != flags[(i + 1) % 4]]
                segs = []
                if len(cut) == 2:
                    i, j = cut
                    # inside corners lie between cut i and cut j (going i+1 .. j) or between cut j and cut i
                    segs.append((e_of(j), e_of(i)) if flags[(i + 1) % 4] else (e_of(i), e_of(j)))
                elif len(cut) == 4:   # ambiguous: cut off each inside corner separately
                    segs += [(e_of(i), e_of(i - 1)) for i in range(4) if flags[i]]
                for a, b in segs:     # directed with the inside on the left for a CCW face
                    if not ccw:
                        a, b = b, a
                    assert a not in nxt
                    nxt[a] = b
        for e in nxt:
            emask[case] |= 1 << e
        assert sorted(nxt) == sorted(nxt.values())
        seen, t = set(), 0
        for start in sorted(nxt):
            if start in seen:
                continue
            loop, cur = [], start
            while cur not in seen:
                seen.add(cur); loop.append(cur); cur = nxt[cur]
            assert cur == start
            for a, b, c in _triangulate(loop):
                tri[case, t] = (a, c, b); t += 1    # reversed: normals point to phi > iso
        ntri[case] = t
    assert ntri.max() <= MAX_TRI
    return ntri, tri, emask


def marching_cubes_numpy(phi, iso=0.0):
    """Slow reference implementation with the same tables and the same vertex / face order as the HIP kernels:
    vertices ordered by owning grid point (x slowest), then axis; faces by cube (x slowest), then table order."""
    ntri, tri, _ = tables()
    phi = np.asarray(phi, np.float64)
    nx, ny, nz = phi.shape
    inside = phi < iso
    vid = -np.ones((nx, ny, nz, 3), np.int64)
    verts = []
    for i in range(nx):
        for j in range(ny):
            for k in range(nz):
                for a, (di, dj, dk) in enumerate(((1, 0, 0), (0, 1, 0), (0, 0, 1))):
                    i2, j2, k2 = i + di, j + dj, k + dk
                    if i2 < nx and j2 < ny and k2 < nz and inside[i, j, k] != inside[i2, j2, k2]:
                        f0, f1 = phi[i, j, k], phi[i2, j2, k2]
                        t = (iso - f0) / (f1 - f0)
                        p = np.array([i, j, k], float); p[a] += t
                        vid[i, j, k, a] = len(verts); verts.append(p)
    faces = []
    for i in range(nx - 1):
        for j in range(ny - 1):
            for k in range(nz - 1):
                case = 0
                for c in range(8):
                    if inside[i + (c & 1), j + ((c >> 1) & 1), k + (c >> 2)]:
                        case |= 1 << c
                for t in range(ntri[case]):
                    f = []
                    for e in tri[case, t]:
                        c0, _ = edge_corners(int(e))
                        f.append(vid[i + c0[0], j + c0[1], k + c0[2], int(e) // 4])
                    faces.append(f)
    return np.array(verts).reshape(-1, 3), np.array(faces, np.int64).reshape(-1, 3)


def marching_cubes_numpy_vec(phi, iso=0.0):
    """`marching_cubes_numpy` with array operations (same vertex / face order); used where a 128^3 grid has to be meshed
    on the host (golden generation)."""
    ntri, tri, _ = tables()
    phi = np.asarray(phi, np.float64)
    nx, ny, nz = phi.shape
    inside = phi < iso
    cut = np.zeros((nx, ny, nz, 3), bool)
    tpar = np.zeros((nx, ny, nz, 3))
    for a, sl_lo, sl_hi in ((0, np.s_[:-1, :, :], np.s_[1:, :, :]), (1, np.s_[:, :-1, :], np.s_[:, 1:, :]),
                            (2, np.s_[:, :, :-1], np.s_[:, :, 1:])):
        c = inside[sl_lo] != inside[sl_hi]
        f0, f1 = phi[sl_lo], phi[sl_hi]
        with np.errstate(divide="ignore", invalid="ignore"):
            t = np.where(c, (iso - f0) / (f1 - f0), 0.0)
        cut[sl_lo + (a,)] = c
        tpar[sl_lo + (a,)] = t
    vid = np.cumsum(cut.reshape(-1)).reshape(cut.shape) - 1
    gi, gj, gk, ga = np.nonzero(cut)
    verts = np.stack([gi, gj, gk], 1).astype(np.float64)
    verts[np.arange(len(ga)), ga] += tpar[gi, gj, gk, ga]
    case = np.zeros((nx - 1, ny - 1, nz - 1), np.int64)
    for c in range(8):
        case |= inside[(c & 1):nx - 1 + (c & 1), ((c >> 1) & 1):ny - 1 + ((c >> 1) & 1), (c >> 2):nz - 1 + (c >> 2)].astype(np.int64) << c
    lin = np.arange(case.size).reshape(case.shape)
    c0 = np.array([edge_corners(e)[0] for e in range(12)], np.int64)
    faces, keys = [], []
    for t in range(MAX_TRI):
        m = ntri[case] > t
        if not m.any():
            break
        ci, cj, ck = np.nonzero(m)
        e = tri[case[m], t].astype(np.int64)                       # [n, 3] edge ids
        f = vid[ci[:, None] + c0[e, 0], cj[:, None] + c0[e, 1], ck[:, None] + c0[e, 2], e // 4]
        faces.append(f)
        keys.append(lin[m] * MAX_TRI + t)
    if not faces:
        return verts.reshape(-1, 3), np.zeros((0, 3), np.int64)
    faces, keys = np.concatenate(faces), np.concatenate(keys)
    return verts, faces[np.argsort(keys, kind="stable")].astype(np.int64)
