"""CPU: csrc/sdf_mesh.hip through the emulator against golden vectors generated from the reference
(`SDF3D.query_sdfs` bodies.py:721-760, `get_ang_inertia` bodies.py:260-395; oracle/gen/gen_sdf_golden.py)."""
import os

import numpy as np
import pytest

from emu import emu

G = os.path.join(os.path.dirname(__file__), "golden")
TYPES = {"box": 0, "sphere": 1, "cylinder": 2, "rounded": 3, "brick": 4, "bowl": 5}


@pytest.mark.parametrize("name", ["box", "sphere", "cylinder", "rounded", "brick", "bowl"])
def test_query_sdfs_matches_reference(name):
    g = np.load(os.path.join(G, "sdf_query.npz"))
    sdf, grad, mask = emu.sdf_query(TYPES[name], g[name + "_prm"], g[name + "_pts"])
    assert np.array_equal(mask, g[name + "_mask"])
    assert np.abs(sdf - g[name + "_sdf"]).max() < 1e-14
    assert np.abs(grad - g[name + "_grad"]).max() < 1e-13
    assert np.all(sdf[~mask] == g[name + "_scale"]) and np.all(grad[~mask] == 0)   # outside the query cube


@pytest.mark.parametrize("name", ["box", "sphere", "cylinder"])
def test_mesh_inertia_matches_reference(name):
    g = np.load(os.path.join(G, "mesh_inertia.npz"))
    J, vol = emu.mesh_inertia(g[name + "_verts"], g[name + "_faces"], float(g[name + "_mass"]))
    assert np.abs(J - g[name + "_J"]).max() < 1e-11 * np.abs(g[name + "_J"]).max()
    assert vol > 0


def test_mesh_inertia_backward_matches_finite_differences():
    """Adjoint of get_ang_inertia w.r.t. the vertices (the reference uses autograd, bodies.py:380-395)."""
    from diffsdfsim_amd import meshes
    v, f = meshes.icosphere(1)
    v = v * np.array([0.7, 0.5, 0.9]) + np.array([0.05, -0.02, 0.03])
    r = np.random.default_rng(2)
    gJ = r.standard_normal((3, 3))
    g = emu.mesh_inertia_backward(v, f, 1.7, gJ)
    h = 1e-6
    for (i, d) in [(0, 0), (5, 1), (11, 2), (20, 0)]:
        vp, vm = v.copy(), v.copy()
        vp[i, d] += h; vm[i, d] -= h
        fd = ((emu.mesh_inertia(vp, f, 1.7)[0] - emu.mesh_inertia(vm, f, 1.7)[0]) * gJ).sum() / (2 * h)
        assert abs(g[i, d] - fd) < 1e-6 * max(1.0, abs(fd)), (i, d, g[i, d], fd)


def test_grid_sdf_query_matches_reference():
    """SDFGrid3D.query_sdfs (bodies.py:203-241, 763-775) recorded from the reference with trilinear `grid_interp`."""
    g = np.load(os.path.join(G, "sdf_query.npz"))
    sdf, grad, mask = emu.grid_sdf_query(g["grid_grid"], float(g["grid_scale"]), g["grid_pts"])
    assert np.array_equal(mask, g["grid_mask"])
    assert np.abs(sdf - g["grid_sdf"]).max() < 1e-14
    assert np.abs(grad - g["grid_grad"]).max() < 1e-12
