"""CPU (emulator): the thinning stage's hull of a big contact cluster (diffsdfsim_amd/csrc/np_common.h: cluster_hull -> hull3_wrap,
gift wrapping) against Qhull (scipy.spatial.ConvexHull, what the reference calls, contacts.py:126-152) on point sets of the kind
a level-set body at rest produces: thousands of points, most of them inside or on flat facets, duplicates, rows of collinear
points -- the vertex SET must be Qhull's."""
import os
import subprocess

import numpy as np
import pytest
from scipy.spatial import ConvexHull

HERE = os.path.dirname(os.path.abspath(__file__))


def run_hull(pts, wave=False):
    out = os.path.join(HERE, "emu", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "check_hull3")
    src = os.path.join(HERE, "emu", "check_hull3.cpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(HERE, "..", "diffsdfsim_amd", "csrc", "np_common.h"))):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-w", "-I", os.path.join(HERE, "emu"), "-o", exe, src])
    f = os.path.join(out, "hull_pts.bin")
    np.ascontiguousarray(pts, np.float64).tofile(f)
    r = subprocess.run([exe, f, str(len(pts))] + (["wave"] if wave else []), capture_output=True, text=True, check=True)
    return np.array(sorted(int(x) for x in r.stdout.split()), np.int64)


def qhull_vertices(pts):
    """Qhull's vertex list, one representative (the lowest index) per group of coincident points."""
    v = ConvexHull(pts).vertices
    keep = set()
    for i in v:
        same = np.nonzero(np.abs(pts - pts[i]).max(axis=1) < 1e-12)[0]
        keep.add(int(same.min()))
    return np.array(sorted(keep), np.int64)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_gift_wrapped_hull_of_a_big_cluster_is_qhulls_vertex_set(seed):
    r = np.random.default_rng(seed)
    # a flat, slightly domed patch like a rounded box resting on its side: a rectangle of coplanar grid points (the resting
    # face), rim rows a fraction of a millimetre above it on a convex profile, interior points, exact duplicates
    nx, ny = 60, 40
    gx, gy = np.meshgrid(np.linspace(-0.3, 0.3, nx), np.linspace(-0.2, 0.2, ny), indexing="ij")
    face = np.stack([gx.ravel(), np.zeros(nx * ny), gy.ravel()], axis=1)
    rim = []
    for k, (h, grow) in enumerate([(1e-5, 0.004), (6e-5, 0.009), (2e-4, 0.016), (5e-4, 0.026)]):
        t = np.linspace(0, 2 * np.pi, 14 + 4 * k, endpoint=False) + 0.01 * seed      # (a few dozen hull vertices, like the real case)
        # a superellipse-like convex ring that widens with height
        rim.append(np.stack([(0.3 + grow) * np.sign(np.cos(t)) * np.abs(np.cos(t)) ** 0.5, np.full_like(t, h),
                             (0.2 + grow) * np.sign(np.sin(t)) * np.abs(np.sin(t)) ** 0.5], axis=1))
    pts = np.concatenate([face] + rim + [face[r.integers(0, len(face), 200)]])          # + duplicates of face points
    q, _ = np.linalg.qr(r.standard_normal((3, 3)))                                       # an arbitrary orientation in the world
    pts = pts @ q.T + r.standard_normal(3)
    pts = pts[r.permutation(len(pts))]
    assert len(pts) > 2048          # beyond the duplicate search: cluster_hull goes to the gift wrap
    mine, ref = run_hull(pts), qhull_vertices(pts)
    assert len(ref) > 20
    assert np.array_equal(mine, ref), (len(mine), len(ref), sorted(set(mine) ^ set(ref))[:10])


def _place(pts, r):
    q, _ = np.linalg.qr(r.standard_normal((3, 3)))
    pts = pts @ q.T + r.standard_normal(3)
    return pts[r.permutation(len(pts))]


def test_gift_wrapped_hull_of_a_box_shaped_cluster():
    """Six big coplanar facets, twelve edges full of collinear points, interior points: eight vertices."""
    r = np.random.default_rng(5)
    g = np.linspace(-1.0, 1.0, 21)
    a, b = np.meshgrid(g, g, indexing="ij")
    one = np.ones(a.size)
    faces = [np.stack(x, axis=1) for x in ((a.ravel(), b.ravel(), one), (a.ravel(), b.ravel(), -one), (a.ravel(), one, b.ravel()),
                                           (a.ravel(), -one, b.ravel()), (one, a.ravel(), b.ravel()), (-one, a.ravel(), b.ravel()))]
    pts = np.concatenate(faces + [0.9 * (2 * r.random((300, 3)) - 1)]) * np.array([0.4, 0.25, 0.3])
    pts = _place(pts, r)
    assert len(pts) > 2048
    mine, ref = run_hull(pts), qhull_vertices(pts)
    assert len(ref) == 8 and np.array_equal(mine, ref), (mine, ref)


def test_gift_wrapped_hull_of_rows_on_a_convex_arc():
    """A cylinder lying in line contact: rows of collinear points along its length on a convex arc -- the ends of every row."""
    r = np.random.default_rng(9)
    th = np.linspace(-0.35, 0.35, 36)
    x = np.linspace(-0.5, 0.5, 70)
    T, X = np.meshgrid(th, x, indexing="ij")
    pts = np.stack([X.ravel(), 0.25 * (1 - np.cos(T.ravel())), 0.25 * np.sin(T.ravel())], axis=1)
    pts = _place(pts, r)
    assert len(pts) > 2048
    mine, ref = run_hull(pts), qhull_vertices(pts)
    assert len(ref) == 72 and np.array_equal(mine, ref), (len(mine), len(ref))


@pytest.mark.parametrize("seed", [11, 12])
def test_gift_wrapped_hull_of_points_in_general_position_with_duplicated_vertices(seed):
    """No coplanar facets at all: 70 points on a sphere (every one a vertex), each present twice, around 2 400 interior points."""
    r = np.random.default_rng(seed)
    v = r.standard_normal((70, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    inner = r.standard_normal((2400, 3)); inner *= (0.6 * r.random((2400, 1)) ** (1 / 3)) / np.linalg.norm(inner, axis=1, keepdims=True)
    pts = np.concatenate([v, inner, v]) * np.array([0.5, 0.3, 0.4])
    pts = _place(pts, r)
    mine, ref = run_hull(pts), qhull_vertices(pts)
    assert len(ref) == 70 and np.array_equal(mine, ref), (len(mine), len(ref))


def reference_flat_vertices(pts):
    """contacts.py:126-152 for a flat cluster: Qhull rejects the 3-D input, the coordinate of least variance is dropped and the
    2-D hull's vertices are kept (one representative per group of coincident points)."""
    drop = int(np.argmin(pts.var(axis=0, ddof=1)))
    p2 = np.delete(pts, drop, axis=1)
    v = ConvexHull(p2).vertices
    keep = set()
    for i in v:
        keep.add(int(np.nonzero(np.abs(pts - pts[i]).max(axis=1) < 1e-12)[0].min()))
    return np.array(sorted(keep), np.int64)


@pytest.mark.parametrize("seed", [21, 22, 23, 24])
def test_flat_cluster_hull_is_the_references_2d_hull(seed):
    """The common case (a box face on a box face): a few hundred coplanar contact points, rows of them collinear along the
    patch's edges up to rotation round-off -- the corners of the patch, nothing on the edges, nothing inside."""
    r = np.random.default_rng(seed)
    nx, ny = int(r.integers(8, 20)), int(r.integers(8, 20))
    gx, gy = np.meshgrid(np.linspace(-0.45, 0.5, nx), np.linspace(-0.3, 0.35, ny), indexing="ij")
    pts = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], axis=1)
    if seed % 2:       # an octagonal patch: the overlap of two rectangles turned against each other
        c, s_ = np.cos(0.3), np.sin(0.3)
        rot = pts[:, :2] @ np.array([[c, -s_], [s_, c]])
        pts = pts[(np.abs(rot[:, 0]) < 0.42) & (np.abs(rot[:, 1]) < 0.3)]
        pts = np.concatenate([pts, pts[r.integers(0, len(pts), 30)]])      # + duplicates
    pts = _place(pts, r)
    mine, ref = run_hull(pts), reference_flat_vertices(pts)
    assert np.array_equal(mine, ref), (mine, ref)
    if len(pts) <= 384:      # the wavefront flavour (what config 3's items run) must agree
        assert np.array_equal(run_hull(pts, wave=True), ref)


@pytest.mark.parametrize("seed", [31, 32, 33])
def test_small_3d_cluster_hull_in_both_flavours(seed):
    """Up to 48 distinct points: the brute-force hull (supporting planes over all triples), workgroup and wavefront flavour."""
    r = np.random.default_rng(seed)
    v = r.standard_normal((14, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    inner = 0.5 * (2 * r.random((25, 3)) - 1) / np.sqrt(3)
    pts = np.concatenate([v, inner, v[:5], 0.5 * (v[0] + v[1])[None]]) * np.array([0.3, 0.2, 0.25])      # duplicates, a mid-edge point
    pts = _place(pts, r)
    ref = qhull_vertices(pts)
    assert np.array_equal(run_hull(pts), ref) and np.array_equal(run_hull(pts, wave=True), ref)
