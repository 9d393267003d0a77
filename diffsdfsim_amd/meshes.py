"""Analytic surface meshes for the primitive SDF bodies (host side, built once per body).

These are the ``custom_mesh=True`` meshes of the reference
(`sdf_physics/physics3d/bodies.py:799-854` box, `:996-1009` sphere, `:939-976` cylinder, `:1029-1065` bowl):
the contact search walks the triangles of body 1's mesh against body 2's SDF, so the
mesh defines the candidate set.  The reference's default, level-set meshes by marching
cubes, is in ``meshsdf.py`` (device kernels).

All functions return ``(verts [V,3] float64, faces [F,3] int64)`` as numpy arrays in the
body frame; ``box_mesh`` additionally returns, per vertex and axis, the sign (+1/-1/0) of
the half-extent that vertex coordinate is tied to -- the reference keeps autograd
connectivity only for the boundary coordinates of the linspace (`bodies.py:807-813`), so
d(vertex)/d(dims) = sign/2 on tied coordinates and 0 elsewhere.
"""
import math

import numpy as np


def _grid_faces(inds):
    a = np.stack([inds[:-1, :-1], inds[1:, :-1], inds[:-1, 1:]], axis=2).reshape(-1, 3)
    b = np.stack([inds[1:, :-1], inds[1:, 1:], inds[:-1, 1:]], axis=2).reshape(-1, 3)
    return np.concatenate([a, b])


def box_mesh(dims, max_tri_length=0.1, grid_axes=None):
    """Six regular grids, one per face, vertices duplicated along box edges.
    ``grid_axes``: the three grid axes (w, h, d) to use instead of this module's linspace (parity tests pass the reference's).

    Follows the layout of `bodies.py:799-854` (front/back, left/right, top/bottom blocks,
    grid pitch <= ``max_tri_length``) so that face ids are comparable with the reference.
    """
    dims = np.asarray(dims, dtype=np.float64)
    hd = dims / 2
    nv = np.ceil(dims / max_tri_length).astype(np.int64) + 1
    axes = []
    for k in range(3):
        # the reference builds the interior with torch.linspace and re-ties both ends to
        # +-half_dims (`bodies.py:803-813`).  torch's CPU linspace is a two-sided fill whose
        # last bit depends on the host's SIMD width, so interior coordinates agree with it
        # to 1 ulp, not bit for bit (parity tests feed the reference's own mesh instead).
        n = int(nv[k])
        lin = _linspace(-hd[k], hd[k], n) if grid_axes is None else np.array(grid_axes[k], np.float64)
        assert len(lin) == n
        lin[0] = -hd[k]
        lin[-1] = hd[k]
        axes.append(lin)
    w, h, d = axes

    def mesh2(a, b):
        A, B = np.meshgrid(a, b, indexing="ij")
        return np.stack([A, B], axis=2).reshape(-1, 2)

    fb, lr, tb = mesh2(w, h), mesh2(h, d), mesh2(w, d)
    fb_i = np.arange(fb.shape[0]).reshape(nv[0], nv[1])
    lr_i = np.arange(lr.shape[0]).reshape(nv[1], nv[2])
    tb_i = np.arange(tb.shape[0]).reshape(nv[0], nv[2])
    f_faces, l_faces, t_faces = _grid_faces(fb_i), _grid_faces(lr_i), _grid_faces(tb_i)

    one = lambda n: np.ones((n, 1))
    f = np.concatenate([fb, hd[2] * one(len(fb))], 1)
    ba = np.concatenate([fb, -hd[2] * one(len(fb))], 1)
    l = np.concatenate([hd[0] * one(len(lr)), lr], 1)
    r = np.concatenate([-hd[0] * one(len(lr)), lr], 1)
    t = np.stack([tb[:, 0], hd[1] * np.ones(len(tb)), tb[:, 1]], 1)
    bo = np.stack([tb[:, 0], -hd[1] * np.ones(len(tb)), tb[:, 1]], 1)
    verts = np.concatenate([f, ba, l, r, t, bo])

    nf, nl, nt = len(f), len(l), len(t)
    faces = np.concatenate([
        f_faces, f_faces[:, ::-1] + nf,
        l_faces + 2 * nf, l_faces[:, ::-1] + 2 * nf + nl,
        t_faces[:, ::-1] + 2 * nf + 2 * nl,
        t_faces + 2 * nf + 2 * nl + nt,
    ]).astype(np.int64)

    # tie[v, k] = +1 / -1 if coordinate k of vertex v *is* +-half_dims[k] with gradient, else 0
    tie = np.zeros_like(verts)
    for k in range(3):
        tie[:, k] = np.where(verts[:, k] == hd[k], 1.0, np.where(verts[:, k] == -hd[k], -1.0, 0.0))
    return verts, faces, tie


def _linspace(start, end, steps):
    """Two-sided float64 linspace (first half from ``start``, second half from ``end``)."""
    if steps == 1:
        return np.array([start], dtype=np.float64)
    step = (end - start) / (steps - 1)
    half = steps // 2
    out = np.empty(steps, dtype=np.float64)
    idx = np.arange(steps)
    out[:half] = start + step * idx[:half]
    out[half:] = end - step * (steps - 1 - idx[half:])
    return out


def icosphere(subdivisions=4):
    """Unit icosphere by recursive midpoint subdivision: 10*4^s + 2 verts, 20*4^s faces.

    Stand-in for ``trimesh.creation.icosphere`` (`bodies.py:1001`); vertex order is
    defined by this function ("vertex order is stand-in-defined").
    """
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array([
        [-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0],
        [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
        [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([
        [0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11],
        [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
        [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9],
        [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    verts = [tuple(x) for x in v]
    for _ in range(subdivisions):
        cache = {}
        new_faces = []

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                m = (np.asarray(verts[a]) + np.asarray(verts[b])) / 2.0
                m /= np.linalg.norm(m)
                verts.append(tuple(m))
                cache[key] = len(verts) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            new_faces += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        f = np.array(new_faces, dtype=np.int64)
    return np.array(verts, dtype=np.float64), f


def sphere_mesh(rad, subdivisions=4):
    v, f = icosphere(subdivisions)
    return v * float(rad), f


def cylinder_mesh(rad, height, numsegs=32, max_tri_length=0.1):
    """Side rings + two fans (layout of `bodies.py:939-976`); axis = z.  Returns verts, faces and the vertex
    gradient rows (d v_x/d rad, d v_y/d rad, d v_z/d height): the side coordinates scale with rad, only the end
    rings and the two cap centres are tied to +-height/2."""
    rad, height = float(rad), float(height)
    thetas = _linspace(0.0, 2 * math.pi * (numsegs - 1) / numsegs, numsegs)
    hh = height / 2
    nv = int(math.ceil(height / max_tri_length)) + 1
    zs = _linspace(-hh, hh, nv)
    zs[0], zs[-1] = -hh, hh
    T, Z = np.meshgrid(thetas, zs, indexing="ij")
    side = np.stack([rad * np.cos(T), rad * np.sin(T), Z], axis=0).reshape(3, -1).T
    verts = np.concatenate([side, [[0.0, 0.0, hh], [0.0, 0.0, -hh]]])
    inds = np.arange(numsegs * nv).reshape(numsegs, nv)
    inds = np.concatenate([inds, inds[:1]], axis=0)
    faces = np.concatenate([
        np.stack([inds[:-1, :-1], inds[1:, 1:], inds[:-1, 1:]], axis=2).reshape(-1, 3),
        np.stack([inds[:-1, :-1], inds[1:, :-1], inds[1:, 1:]], axis=2).reshape(-1, 3)])
    V = len(verts)
    top = np.stack([np.full(numsegs, V - 2), inds[:-1, -1], inds[1:, -1]], axis=1)
    bot = np.stack([np.full(numsegs, V - 1), inds[1:, 0], inds[:-1, 0]], axis=1)
    faces = np.concatenate([faces, top, bot]).astype(np.int64)
    vgrad = np.zeros_like(verts)
    vgrad[:len(side), 0] = np.cos(T).reshape(-1)
    vgrad[:len(side), 1] = np.sin(T).reshape(-1)
    vgrad[:, 2] = np.where(verts[:, 2] == hh, 0.5, np.where(verts[:, 2] == -hh, -0.5, 0.0))
    return verts, faces, vgrad


def bowl_mesh(r, d, numsegs=32):
    """Hemispherical shell of mid radius ``r`` and half thickness ``d``, opening towards +z, rim closed by a flat ring
    (layout of `bodies.py:1029-1065`: inner and outer hemisphere of numsegs/4 latitude rings x numsegs meridians, the pole
    ring collapsed to one vertex; all vertices lifted by r/2)."""
    r, d = float(r), float(d)
    nt = numsegs // 4
    thetas = _linspace(0.0, -math.pi / 2, nt)
    phis = _linspace(0.0, 2 * math.pi * (numsegs - 1) / numsegs, numsegs)
    T, P = np.meshgrid(thetas, phis, indexing="ij")

    def sph(rad):
        rc = rad * np.cos(T)
        return np.stack([rc * np.cos(P), rc * np.sin(P), rad * np.sin(T)], axis=0).reshape(3, -1).T

    inds = np.arange(numsegs * nt).reshape(nt, numsegs)
    n0 = int(inds[-1, 1])                       # keep one vertex of the pole ring
    v0, v1 = sph(r - d)[:n0], sph(r + d)[:n0]
    inds[-1] = inds[-1, 0]
    inds = np.concatenate([inds, inds[:, :1]], axis=1)
    shell = np.concatenate([np.stack([inds[1:, 1:], inds[:-1, 1:], inds[:-1, :-1]], axis=2).reshape(-1, 3),
                            np.stack([inds[1:-1, :-1], inds[1:-1, 1:], inds[:-2, :-1]], axis=2).reshape(-1, 3)])
    rim = np.concatenate([np.stack([inds[0, :-1] + n0, inds[0, 1:] + n0, inds[0, :-1]], axis=1),
                          np.stack([inds[0, 1:] + n0, inds[0, 1:], inds[0, :-1]], axis=1)])
    verts = np.concatenate([v0, v1])
    faces = np.concatenate([shell[:, ::-1], shell + n0, rim]).astype(np.int64)
    verts[:, 2] += r / 2
    return verts, faces
