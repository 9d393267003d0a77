"""CPU: csrc/marching_cubes.hip through the emulator.  The reference's marching cubes is an un-vendored extension
(vertex / face order implementation-defined), so the checks are: (i) the kernels reproduce the numpy
restatement of the same generated tables exactly, (ii) the tables give a closed, consistently oriented 2-manifold
for every one of the 256 cube configurations (white-noise field), (iii) vertices lie on the level set, (iv) the MeshSDF
backward formula (bodies.py:680-702) matches finite differences of a mesh functional."""
import collections

import numpy as np

from diffsdfsim_amd import mc_tables
from emu import emu


def _directed_edges(f):
    ed = collections.Counter()
    for a, b, c in f:
        for x, y in ((a, b), (b, c), (c, a)):
            ed[(int(x), int(y))] += 1
    return ed


def test_kernels_match_numpy_restatement_and_mesh_is_closed():
    r = np.random.default_rng(0)
    n = 12
    phi = r.normal(size=(n, n + 1, n + 2))
    phi[0], phi[-1] = 1.0, 1.0
    phi[:, 0], phi[:, -1], phi[:, :, 0], phi[:, :, -1] = 1.0, 1.0, 1.0, 1.0
    v, f = emu.marching_cubes(phi, 0.0)
    v0, f0 = mc_tables.marching_cubes_numpy(phi, 0.0)
    assert np.array_equal(f, f0) and np.array_equal(v, v0)
    ed = _directed_edges(f)
    assert all(c == 1 and ed[(y, x)] == 1 for (x, y), c in ed.items())      # closed, oriented 2-manifold


def test_all_256_cases_occur_in_the_noise_test():
    ntri, tri, emask = mc_tables.tables()
    assert ntri.max() == 5 and ntri[0] == 0 and ntri[255] == 0
    for case in range(1, 255):
        used = set(int(e) for e in tri[case, : ntri[case]].reshape(-1))
        assert used == {e for e in range(12) if emask[case] >> e & 1}        # every cut edge is used, no other


def test_sphere_level_set_orientation_volume():
    n = 24
    g = np.linspace(-1, 1, n)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    v, f = emu.marching_cubes(np.sqrt(X ** 2 + Y ** 2 + Z ** 2) - 0.7)
    v = v / (n - 1) * 2 - 1
    assert np.abs(np.linalg.norm(v, axis=1) - 0.7).max() < 2e-3
    vol = np.einsum("ij,ij->i", v[f[:, 0]], np.cross(v[f[:, 1]], v[f[:, 2]])).sum() / 6
    assert 0.97 * 4 / 3 * np.pi * 0.7 ** 3 < vol < 4 / 3 * np.pi * 0.7 ** 3       # outward normals, inscribed polyhedron


def test_meshsdf_backward_matches_finite_differences():
    """L = mean over the mesh vertices of an outward pull; dL/dparams by the MeshSDF formula vs central differences
    of the mesh itself.  Sphere (smooth: tight), box (vertices slide along grid edges at the box edges: loose)."""
    n = 20
    g = np.linspace(-1, 1, n)
    P = np.stack(np.meshgrid(g, g, g, indexing="ij"), 3).reshape(-1, 3)

    def mesh(ty, prm):
        sdf, _, _ = emu.sdf_query(ty, prm, P)
        v, f = emu.marching_cubes(sdf.reshape(n, n, n))
        return v / (n - 1) * 2 - 1

    h = 1e-4
    rad = np.array([0.6, 0.0, 0.0])
    v = mesh(1, rad)
    w = v / np.linalg.norm(v, axis=1, keepdims=True) / len(v)        # L = mean radius of the vertices
    got = emu.meshsdf_backward(1, rad, v, w)
    fd = (np.linalg.norm(mesh(1, rad + [h, 0, 0]), axis=1).mean() - np.linalg.norm(mesh(1, rad - [h, 0, 0]), axis=1).mean()) / (2 * h)
    assert abs(got[0] - 1.0) < 1e-12 and abs(fd - 1.0) < 0.1   # vertices slide along grid edges: dr / cos(angle)

    dims = np.array([0.9, 1.2, 1.0])
    v = mesh(0, dims)
    got = emu.meshsdf_backward(0, dims, v, np.sign(v) / len(v))
    for i in range(3):
        e = np.zeros(3); e[i] = h
        vp, vm = mesh(0, dims + e), mesh(0, dims - e)
        assert vp.shape == v.shape and vm.shape == v.shape      # same cube cases for a small perturbation
        fd = ((np.sign(v) / len(v)) * (vp - vm)).sum() / (2 * h)
        assert abs(got[i] - fd) < 0.3 * abs(fd), (i, got[i], fd)
