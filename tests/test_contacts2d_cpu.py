"""CPU: the 2-D analytic contact kernels (csrc/contacts2d.hip), compiled for the test-only emulator,
against golden vectors recorded from the reference's DiffContactHandler (oracle/gen/gen_contacts2d_golden.py): 400 pairs of
circles and convex polygons -- contact tuples, the separating-axis state left on the bodies, and autograd's gradients."""
import os

import numpy as np

from emu import emu
from helpers import GOLDEN


def load():
    g = np.load(os.path.join(GOLDEN, "contacts2d.npz"))
    sw = lambda a: np.ascontiguousarray(np.swapaxes(a, 0, 1))      # noqa: E731  [P][2]... -> [2][P]...
    return g, dict(kind=sw(g["kind"]), nv=sw(g["nv"]), pos=sw(g["pos"]), rad=sw(g["rad"]), verts=sw(g["verts"]),
                   sat_in=sw(g["sat_in"]), eps=float(g["eps"]))


def test_contact_tuples_and_axis_state_match_the_reference():
    g, a = load()
    out, count, sat_out = emu.contacts2d_forward(**a)
    assert (count == g["count"]).all()
    assert (sat_out.T == g["sat_out"]).all()
    assert np.abs(out - g["out"]).max() < 1e-12
    assert set(np.unique(count)) == {0, 1, 2}


def test_gradients_match_reference_autograd():
    g, a = load()
    g_pos, g_rad, g_verts = emu.contacts2d_backward(gout=g["gout"], **a)
    for mine, ref in ((g_pos, g["g_pos"]), (g_rad, g["g_rad"]), (g_verts, g["g_verts"])):
        ref = np.swapaxes(ref, 0, 1)
        assert np.abs(ref).max() > 0.1
        assert np.abs(mine - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())


def test_every_branch_of_the_handler_is_in_the_golden_set():
    g, _ = load()
    kinds = [tuple(k) for k in g["kind"]]
    for pair in ((0, 0), (0, 1), (1, 0), (1, 1)):
        m = np.array([k == pair for k in kinds])
        assert (g["count"][m] > 0).sum() >= 20 and (g["count"][m] == 0).sum() >= 10
    # circle inside a polygon (the separating-axis branch): the contact point lies more than the radius from the centre
    m = np.array([k in ((0, 1), (1, 0)) for k in kinds]) & (g["count"] == 1)
    deep = [i for i in np.nonzero(m)[0] if g["out"][i, 0, 6] > g["rad"][i].max() + 1e-9]
    assert len(deep) >= 5


def test_numpy_restatement_with_the_gjk_walk_is_pinned_by_the_same_goldens():
    """oracle/contacts2d_oracle.py follows the reference's walk (GJK from vertex 0) where the kernels search edge by edge; both
    must land on what the reference recorded."""
    from oracle import contacts2d_oracle as O
    g, a = load()
    out, count, sat_out = O.contacts2d(**a)
    assert (count == g["count"]).all() and (sat_out.T == g["sat_out"]).all()
    assert np.abs(out - g["out"]).max() < 1e-11


def test_kernels_and_restatement_agree_on_fresh_random_pairs():
    from oracle import contacts2d_oracle as O
    a = O.random_pairs(np.random.default_rng(2024), 300)
    ref = O.contacts2d(eps=0.1, **a)
    got = emu.contacts2d_forward(eps=0.1, **a)
    assert (got[1] == ref[1]).all() and (got[2] == ref[2]).all()
    assert np.abs(got[0] - ref[0]).max() < 1e-11
    assert (ref[1] == 2).sum() > 10 and (ref[1] == 0).sum() > 30
