"""GPU: dss_sdf_query / dss_mesh_inertia through the C ABI against golden vectors generated from the reference
(`SDF3D.query_sdfs` bodies.py:721-760, `get_ang_inertia` bodies.py:260-395).  Tolerances: sdf 1e-14, grad 1e-13
(same IEEE operations in the same order), inertia 1e-11 relative (the reduction order over faces differs)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
TYPES = {"box": 0, "sphere": 1, "cylinder": 2}


@pytest.mark.parametrize("name", ["box", "sphere", "cylinder"])
def test_query_sdfs_matches_reference(name):
    from diffsdfsim_amd.mass_properties import sdf_query
    g = np.load(os.path.join(G, "sdf_query.npz"))
    sdf, grad, mask = sdf_query(TYPES[name], g[name + "_prm"], g[name + "_pts"], True, True)
    sdf, grad, mask = sdf.cpu().numpy(), grad.cpu().numpy(), mask.cpu().numpy()
    assert np.array_equal(mask, g[name + "_mask"])
    assert np.abs(sdf - g[name + "_sdf"]).max() < 1e-14
    assert np.abs(grad - g[name + "_grad"]).max() < 1e-13
    only = sdf_query(TYPES[name], g[name + "_prm"], g[name + "_pts"], False, False)
    assert torch.equal(only.cpu(), torch.as_tensor(sdf))


def test_body_query_sdfs_api():
    from diffsdfsim_amd.physics3d import SDFBox
    g = np.load(os.path.join(G, "sdf_query.npz"))
    b = SDFBox([0, 0, 0], g["box_prm"], custom_mesh=True, custom_inertia=True)
    sdf, grad = b.query_sdfs(torch.as_tensor(g["box_pts"]))
    assert np.abs(sdf.cpu().numpy() - g["box_sdf"]).max() < 1e-14 and np.abs(grad.cpu().numpy() - g["box_grad"]).max() < 1e-13


@pytest.mark.parametrize("name", ["box", "sphere", "cylinder"])
def test_mesh_inertia_matches_reference(name):
    from diffsdfsim_amd.mass_properties import mesh_inertia
    g = np.load(os.path.join(G, "mesh_inertia.npz"))
    J, vol = mesh_inertia(g[name + "_verts"], g[name + "_faces"], float(g[name + "_mass"]), return_volume=True)
    assert np.abs(J.cpu().numpy() - g[name + "_J"]).max() < 1e-11 * np.abs(g[name + "_J"]).max()


def test_mesh_inertia_of_own_meshes_is_close_to_analytic():
    """custom_inertia=False: the body integrates its own mesh (reference: SDF3D._get_ang_inertia, bodies.py:713-714)."""
    from diffsdfsim_amd.physics3d import SDFBox, SDFSphere
    b = SDFBox([0, 0, 0], [0.9, 1.1, 1.3], mass=2.0, custom_mesh=True, custom_inertia=False)
    a = SDFBox([0, 0, 0], [0.9, 1.1, 1.3], mass=2.0, custom_mesh=True, custom_inertia=True)
    assert torch.allclose(b.ang_inertia, a.ang_inertia, atol=1e-12)      # a box mesh is the box
    s = SDFSphere([0, 0, 0], 0.5, mass=2.0, custom_mesh=True, custom_inertia=False)
    t = SDFSphere([0, 0, 0], 0.5, mass=2.0, custom_mesh=True, custom_inertia=True)
    assert torch.allclose(s.ang_inertia, t.ang_inertia, rtol=5e-3)       # icosphere(4) vs the exact sphere


def test_batched_mesh_table():
    from diffsdfsim_amd.mass_properties import mesh_inertia
    g = np.load(os.path.join(G, "mesh_inertia.npz"))
    names = ["box", "sphere", "cylinder"]
    J = mesh_inertia([g[n + "_verts"] for n in names], [g[n + "_faces"] for n in names], [2.5] * 3)
    for k, n in enumerate(names):
        assert np.abs(J[k].cpu().numpy() - g[n + "_J"]).max() < 1e-11 * np.abs(g[n + "_J"]).max()
