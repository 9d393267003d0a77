"""CPU: csrc/sdf_mesh.hip through the emulator against golden vectors generated from the reference
(`SDF3D.query_sdfs` bodies.py:721-760, `get_ang_inertia` bodies.py:260-395; oracle/gen/gen_sdf_golden.py)."""
import os

import numpy as np
import pytest

from emu import emu

G = os.path.join(os.path.dirname(__file__), "golden")
TYPES = {"box": 0, "sphere": 1, "cylinder": 2}


@pytest.mark.parametrize("name", ["box", "sphere", "cylinder"])
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
