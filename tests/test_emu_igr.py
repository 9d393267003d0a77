"""CPU: logic of csrc/igr_mlp.hip (forward-mode tangents in MFMA tiles) through the emulator vs the numpy oracle."""
import numpy as np

from emu import emu
from oracle import igr_oracle as IO


def test_igr_value_and_input_gradient():
    Ws, bs = IO.geometric_init(seed=3)
    r = np.random.default_rng(1)
    pts = r.uniform(-1, 1, (21, 3))        # not a multiple of the 8-point tile
    lat = r.normal(0, 0.1, 2)
    sdf, grad = emu.igr_query(pts, lat, Ws, bs)
    so, go = IO.query(pts, lat, Ws, bs)
    assert np.abs(sdf - so).max() < 1e-12 and np.abs(grad - go).max() < 1e-11
    # the gradient really is d sdf / d xyz
    h = 1e-6
    for d in range(3):
        e = np.zeros(3); e[d] = h
        fd = (IO.query(pts + e, lat, Ws, bs)[0] - IO.query(pts - e, lat, Ws, bs)[0]) / (2 * h)
        assert np.abs(fd - go[:, d]).max() < 1e-6


def test_igr_latent_gradient():
    Ws, bs = IO.geometric_init(seed=3)
    r = np.random.default_rng(2)
    pts = r.uniform(-1, 1, (13, 3)); lat = r.normal(0, 0.1, 2)
    sdf, gl = emu.igr_query(pts, lat, Ws, bs, wrt="latent")
    so, go = IO.query(pts, lat, Ws, bs, wrt="latent")
    assert np.abs(sdf - so).max() < 1e-12 and np.abs(gl - go).max() < 1e-11 and np.all(gl[:, 2] == 0)
    h = 1e-6
    for d in range(2):
        e = np.zeros(2); e[d] = h
        fd = (IO.query(pts, lat + e, Ws, bs)[0] - IO.query(pts, lat - e, Ws, bs)[0]) / (2 * h)
        assert np.abs(fd - go[:, d]).max() < 1e-6
