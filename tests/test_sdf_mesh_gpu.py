"""GPU: dss_sdf_query / dss_mesh_inertia through the C ABI against golden vectors generated from the reference
(`SDF3D.query_sdfs` bodies.py:721-760, `get_ang_inertia` bodies.py:260-395).  Tolerances: sdf 1e-14, grad 1e-13
(same IEEE operations in the same order), inertia 1e-11 relative (the reduction order over faces differs)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
TYPES = {"box": 0, "sphere": 1, "cylinder": 2, "rounded": 3, "brick": 4, "bowl": 5}


@pytest.mark.parametrize("name", ["box", "sphere", "cylinder", "rounded", "brick", "bowl"])
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


def test_shared_reciprocal_division_is_bit_identical_to_ieee_division():
    """geom.h div3 (one reciprocal refinement for three quotients) vs three `/` on the device, over the magnitudes
    the geometry kernels see (lengths down to the 1e-12 clamp, coordinates up to 1e3) and well beyond."""
    from diffsdfsim_amd import _lib
    r = np.random.default_rng(5)
    n = 1 << 21
    num = r.standard_normal((n, 3)) * 10.0 ** r.uniform(-15, 6, (n, 1))
    den = np.abs(r.standard_normal(n)) * 10.0 ** r.uniform(-13, 6, n) + 1e-300
    num[:1000] = 0.0; den[1000:2000] = 1e-12; den[2000:3000] = 3.0; num[3000:4000] = den[3000:4000, None]
    tn, td = torch.tensor(num, device="cuda"), torch.tensor(den, device="cuda")
    bad = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().dss_selftest_div3(_lib.ptr(tn), _lib.ptr(td), n, _lib.ptr(bad), _lib.stream_ptr(tn.device)), "dss_selftest_div3")
    assert int(bad.item()) == 0


def test_unscaled_square_root_is_bit_identical_to_sqrt():
    from diffsdfsim_amd import _lib
    r = np.random.default_rng(6)
    n = 1 << 22
    x = np.abs(r.standard_normal(n)) * 10.0 ** r.uniform(-40, 12, n)
    x[:1000] = 0.0
    x[1000:2000] = np.arange(1000.0) ** 2
    x[2000:3000] = np.nextafter(np.arange(1.0, 1001.0) ** 2, 0)
    tx = torch.tensor(x, device="cuda")
    bad = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().dss_selftest_sqrt(_lib.ptr(tx), n, _lib.ptr(bad), _lib.stream_ptr(tx.device)), "dss_selftest_sqrt")
    assert int(bad.item()) == 0


def test_grid_body_queries_mesh_and_inertia():
    """SDFGrid3D (bodies.py:763-775): queries against the reference's vector, mesh = level set of the grid (closed,
    vertices on the zero set of the interpolant), inertia from that mesh."""
    from diffsdfsim_amd.physics3d import SDFGrid3D
    g = np.load(os.path.join(G, "sdf_query.npz"))
    b = SDFGrid3D([0, 0, 0], float(g["grid_scale"]), g["grid_grid"], mass=2.0)
    sdf, grad, mask = b.query_sdfs(torch.as_tensor(g["grid_pts"]), True, True)
    assert np.array_equal(mask.cpu().numpy(), g["grid_mask"])
    assert np.abs(sdf.cpu().numpy() - g["grid_sdf"]).max() < 1e-14 and np.abs(grad.cpu().numpy() - g["grid_grad"]).max() < 1e-12
    on_surface = b.query_sdfs(b.verts, return_grads=False)
    assert float(on_surface.abs().max()) < 1e-12           # marching-cubes vertices lie on grid edges: the interpolant is linear there
    f = b.faces.numpy()
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
    _, cnt = np.unique(e, axis=0, return_counts=True)
    assert (cnt == 2).all()                                  # closed manifold
    J = b.ang_inertia.numpy()
    assert np.allclose(J, J.T, atol=1e-12) and (np.linalg.eigvalsh(J) > 0).all()
