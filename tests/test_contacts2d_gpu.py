"""GPU: the 2-D analytic contact handler (csrc/contacts2d.hip through the C ABI and
diffsdfsim_amd.physics2d) against the golden vectors recorded from the reference's DiffContactHandler."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN

pytestmark = pytest.mark.gpu


def load(dev):
    g = np.load(os.path.join(GOLDEN, "contacts2d.npz"))
    sw = lambda a, dt: torch.as_tensor(np.ascontiguousarray(np.swapaxes(a, 0, 1)), dtype=dt, device=dev)      # noqa: E731
    a = dict(pos=sw(g["pos"], torch.float64), rad=sw(g["rad"], torch.float64), verts=sw(g["verts"], torch.float64),
             kind=sw(g["kind"], torch.int32), nv=sw(g["nv"], torch.int32), sat_in=sw(g["sat_in"], torch.int32), eps=float(g["eps"]))
    return g, a


def test_contacts_and_gradients_match_the_reference_handler():
    from diffsdfsim_amd.physics2d import contacts2d
    g, a = load("cuda:0")
    for k in ("pos", "rad", "verts"):
        a[k].requires_grad_(True)
    out, count, sat_out = contacts2d(**a)
    assert (count.cpu().numpy() == g["count"]).all()
    assert (sat_out.cpu().numpy().T == g["sat_out"]).all()
    assert np.abs(out.detach().cpu().numpy() - g["out"]).max() < 1e-12
    (out * torch.as_tensor(g["gout"], device=out.device)).sum().backward()
    for mine, ref in ((a["pos"].grad, g["g_pos"]), (a["rad"].grad, g["g_rad"]), (a["verts"].grad, g["g_verts"])):
        ref = np.swapaxes(ref, 0, 1)
        assert np.abs(mine.cpu().numpy() - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())


def test_one_pair_at_a_time_equals_the_batch():
    from diffsdfsim_amd.physics2d import contacts2d
    g, a = load("cuda:0")
    out, count, sat_out = contacts2d(**a)
    for i in (0, 57, 131, 222, 399):
        one = {k: (v[:, i:i + 1].contiguous() if torch.is_tensor(v) else v) for k, v in a.items()}
        o1, c1, s1 = contacts2d(**one)
        assert torch.equal(o1[0], out[i]) and int(c1[0]) == int(count[i]) and torch.equal(s1[:, 0], sat_out[:, i])


def test_handler_appends_the_reference_tuples_for_world_bodies_on_the_host():
    """The handler as the reference's world calls it (`contact_callback(args, geom1, geom2)`, world.py:396-399), with bodies
    whose tensors live on the host: geometry goes to the device, tuples and gradients come back."""
    from diffsdfsim_amd.physics2d import make_handler
    g = np.load(os.path.join(GOLDEN, "contacts2d.npz"))

    class Obj:
        pass
    for i in list(np.nonzero(g["count"] == 2)[0][:3]) + list(np.nonzero((g["count"] == 1) & (g["kind"][:, 0] == 1) & (g["kind"][:, 1] == 0))[0][:3]):
        bodies, leaves = [], []
        for s in range(2):
            b = Obj()
            b.pos = torch.tensor(g["pos"][i, s], requires_grad=True)
            leaves.append(b.pos)
            if g["kind"][i, s] == 0:
                b.rad = torch.tensor(g["rad"][i, s], requires_grad=True)
                leaves.append(b.rad)
            else:
                b.verts = [torch.tensor(v, requires_grad=True) for v in g["verts"][i, s, :g["nv"][i, s]]]
                b.last_sat_idx = int(g["sat_in"][i, s])
                leaves += b.verts
            bodies.append(b)
        w = Obj(); w.bodies, w.eps, w.contacts = bodies, float(g["eps"]), []
        g1, g2 = Obj(), Obj()
        g1.body, g2.body, g1.no_contact, g2.no_contact = 0, 1, set(), set()
        make_handler(device="cuda:0")()([w], g1, g2)
        assert len(w.contacts) == int(g["count"][i])
        loss = 0.0
        for q, (c, i1, i2) in enumerate(w.contacts):
            assert (i1, i2) == (0, 1) and c[0].device.type == "cpu"
            flat = torch.cat([c[0], c[1], c[2], c[3].reshape(1)])
            assert np.abs(flat.detach().numpy() - g["out"][i, q]).max() < 1e-12
            loss = loss + (flat * torch.tensor(g["gout"][i, q])).sum()
        loss.backward()
        for s, b in enumerate(bodies):
            assert np.abs(b.pos.grad.numpy() - g["g_pos"][i, s]).max() < 1e-9
            if hasattr(b, "verts"):
                assert b.last_sat_idx == int(g["sat_out"][i, s])
                got = np.stack([v.grad.numpy() if v.grad is not None else np.zeros(2) for v in b.verts])
                assert np.abs(got - g["g_verts"][i, s, :len(b.verts)]).max() < 1e-9
            else:
                assert abs(float(b.rad.grad) - g["g_rad"][i, s]) < 1e-9
        g2.no_contact.add(g1)
        w.contacts = []
        make_handler(device="cuda:0")()([w], g1, g2)
        assert w.contacts == []


def test_limits_and_empty_batch():
    from diffsdfsim_amd import _lib
    from diffsdfsim_amd.physics2d import contacts2d
    dev = "cuda:0"
    z = lambda *s, dt=torch.float64: torch.zeros(*s, dtype=dt, device=dev)      # noqa: E731
    out, count, sat = contacts2d(z(2, 0, 2), z(2, 0), z(2, 0, 4, 2), z(2, 0, dt=torch.int32), z(2, 0, dt=torch.int32), z(2, 0, dt=torch.int32), 0.1)
    assert out.shape == (0, 2, 7) and count.numel() == 0
    with pytest.raises(_lib.HipLibraryError):
        contacts2d(z(2, 1, 2), z(2, 1), z(2, 1, 9, 2), z(2, 1, dt=torch.int32), z(2, 1, dt=torch.int32), z(2, 1, dt=torch.int32), 0.1)
    with pytest.raises(_lib.HipLibraryError):
        contacts2d(z(2, 1, 2).cpu(), z(2, 1).cpu(), z(2, 1, 4, 2).cpu(), z(2, 1, dt=torch.int32).cpu(), z(2, 1, dt=torch.int32).cpu(),
                   z(2, 1, dt=torch.int32).cpu(), 0.1)


def _to_dev(a, dev="cuda:0"):
    t = {}
    for k, v in a.items():
        t[k] = torch.as_tensor(v, device=dev, dtype=torch.float64 if v.dtype.kind == "f" else torch.int32).contiguous()
    return t


def test_fresh_random_pairs_against_the_numpy_restatement():
    """4000 pairs no golden covers: the device kernels (closest feature found edge by edge) against the restatement that
    walks GJK as the reference does (oracle/contacts2d_oracle.py, pinned by the goldens)."""
    from diffsdfsim_amd.physics2d import contacts2d
    from oracle import contacts2d_oracle as O
    a = O.random_pairs(np.random.default_rng(99), 4000)
    ref_out, ref_count, ref_sat = O.contacts2d(eps=0.1, **a)
    out, count, sat = contacts2d(eps=0.1, **_to_dev(a))
    assert (count.cpu().numpy() == ref_count).all() and (sat.cpu().numpy() == ref_sat).all()
    assert np.abs(out.cpu().numpy() - ref_out).max() < 1e-11
    assert all((ref_count == c).sum() > 100 for c in (0, 1, 2))


def test_vector_jacobian_product_against_central_differences_of_the_restatement():
    """d <gout, contacts> along a random direction of all coordinates: the backward kernel against central differences of the
    numpy restatement, on the pairs whose feature decisions do not change within the difference step."""
    from diffsdfsim_amd.physics2d import contacts2d
    from oracle import contacts2d_oracle as O
    rng = np.random.default_rng(5)
    a = O.random_pairs(rng, 600)
    t = _to_dev(a)
    for k in ("pos", "rad", "verts"):
        t[k].requires_grad_(True)
    out, count, _ = contacts2d(eps=0.1, **t)
    gout = rng.normal(size=out.shape)
    (out * torch.as_tensor(gout, device=out.device)).sum().backward()
    mask_v = (np.arange(a["verts"].shape[2])[None, None, :, None] < a["nv"][:, :, None, None])
    d = {"pos": rng.normal(size=a["pos"].shape), "rad": rng.normal(size=a["rad"].shape) * (a["kind"] == 0),
         "verts": rng.normal(size=a["verts"].shape) * mask_v}
    ad = sum((t[k].grad.cpu().numpy() * d[k]).reshape(2, 600, -1).sum((0, 2)) for k in d)      # per pair

    def f(h):
        b = dict(a)
        for k in d:
            b[k] = a[k] + h * d[k]
        o, c, s = O.contacts2d(eps=0.1, **b)
        return (o * gout).sum((1, 2)), c, s
    checked = 0
    (fp, cp, sp), (fm, cm, sm) = f(1e-6), f(-1e-6)
    (fp2, cp2, _), (fm2, cm2, _) = f(2e-6), f(-2e-6)
    cnt = count.cpu().numpy()
    for p in range(600):
        if cnt[p] == 0 or not (cp[p] == cm[p] == cp2[p] == cm2[p] == cnt[p]) or not (sp[:, p] == sm[:, p]).all():
            continue
        fd1, fd2 = (fp[p] - fm[p]) / 2e-6, (fp2[p] - fm2[p]) / 4e-6
        if abs(fd1 - fd2) > 1e-5 * max(1.0, abs(fd1)):      # a feature switch inside the step
            continue
        assert abs(fd1 - ad[p]) < 2e-5 * max(1.0, abs(fd1)), (p, fd1, ad[p], a["kind"][:, p])
        checked += 1
    assert checked > 250


def test_config1_end_to_end_on_the_device_library():
    """BASELINE configs[0] without the reference: the 2-D world of diffsdfsim_amd.physics2d (contacts of all pairs from the
    contact kernel, the LCP from the dense kernel, assembly in torch on the device), 50 steps forward + backward, against what
    the reference's own world produced (tests/golden/config1_lcp.npz): number of LCP solves and sub-steps, final poses, loss,
    d loss / d radius."""
    from diffsdfsim_amd.physics2d import Circle, Gravity, Rect, TotalConstraint, World
    g = np.load(os.path.join(GOLDEN, "config1_lcp.npz"))
    rad = torch.tensor(20.0, dtype=torch.double, requires_grad=True)
    floor = Rect([500, 600], [1000, 50], restitution=0.5, fric_coeff=0.9)
    ball = Circle([500, 480], rad, vel=[0, 30, 0], restitution=0.5, fric_coeff=0.9)
    ball.add_force(Gravity(g=100))
    w = World([floor, ball], [TotalConstraint(floor)], dt=1.0 / 30)
    for _ in range(50):
        w.step()
    loss = (ball.pos ** 2).sum()
    loss.backward()
    assert w.lcp_calls == int(g["n_calls"]) and len(w.trajectory) == int(g["n_substeps"])
    assert abs(w.t - float(g["t_final"])) < 1e-12
    final_p = torch.cat([floor.p, ball.p]).detach().cpu().numpy()
    assert np.abs(final_p - g["final_p"]).max() < 1e-8 * np.abs(g["final_p"]).max()
    assert abs(float(loss) - float(g["loss"])) < 1e-8 * abs(float(g["loss"]))
    assert abs(float(rad.grad) - float(g["drad"])) < 1e-6 * abs(float(g["drad"])), (float(rad.grad), float(g["drad"]))


def test_polygon_scene_end_to_end_against_the_reference_world():
    """Polygon against polygon (reference / incident edge and clipping: a tilted box landing on a corner and tipping onto its
    face) and circle against polygon (a ball rolling into the box) in one 2-D world on the device library, 60 steps, against
    the reference's own world (tests/golden/config1_polygons.npz): contact pairs of every step, poses along the way, loss and
    d loss / d (box width, ball radius)."""
    from diffsdfsim_amd.physics2d import Circle, Gravity, Rect, TotalConstraint, World
    g = np.load(os.path.join(GOLDEN, "config1_polygons.npz"))
    wd = torch.tensor(80.0, dtype=torch.double, requires_grad=True)
    rad = torch.tensor(25.0, dtype=torch.double, requires_grad=True)
    floor = Rect([500, 600], [1000, 50], restitution=0.2, fric_coeff=0.6)
    box = Rect([0.3, 420, 520], torch.stack([wd, wd.new_tensor(50.0)]), restitution=0.2, fric_coeff=0.6)
    ball = Circle([560, 500], rad, vel=[0, -150, 0], restitution=0.2, fric_coeff=0.6)
    for b in (box, ball):
        b.add_force(Gravity(g=100))
    w = World([floor, box, ball], [TotalConstraint(floor)], dt=1.0 / 30)
    for k in range(60):
        w.step()
        want = [tuple(r) for r in g["pairs"][k] if r[0] >= 0]
        assert [(c[1], c[2]) for c in w.contacts] == want, (k, want)
        p = torch.cat([b.p for b in (floor, box, ball)]).detach().cpu().numpy()
        assert np.abs(p - g["traj_p"][k]).max() < 1e-7 * np.abs(g["traj_p"][k]).max(), k
    assert len(w.trajectory) == int(g["n_substeps"])
    loss = (box.p ** 2).sum() + (ball.pos ** 2).sum()
    gw, gr = torch.autograd.grad(loss, [wd, rad])
    assert abs(float(loss) - float(g["loss"])) < 1e-8 * abs(float(g["loss"]))
    assert abs(float(gw) - float(g["g_width"])) < 1e-5 * abs(float(g["g_width"])), (float(gw), float(g["g_width"]))
    assert abs(float(gr) - float(g["g_rad"])) < 1e-5 * abs(float(g["g_rad"])), (float(gr), float(g["g_rad"]))


def test_general_hulls_end_to_end_against_the_reference_world():
    """`Hull` bodies that are not rectangles (centroid shift and polygon inertia of bodies.py:196-254): a spinning pentagon,
    scaled by a differentiable factor, and a triangle on the slab, 50 steps, against the reference's world
    (tests/golden/config1_hulls.npz): inertias, contact pairs of every step, poses, loss and d loss / d scale."""
    from diffsdfsim_amd.physics2d import Gravity, Hull, Rect, TotalConstraint, World
    g = np.load(os.path.join(GOLDEN, "config1_hulls.npz"))
    sc = torch.tensor(1.0, dtype=torch.double, requires_grad=True)
    pent = [[40.0, 0.0], [12.0, 38.0], [-32.0, 24.0], [-32.0, -24.0], [12.0, -38.0]]
    tri = [[30.0, 20.0], [-30.0, 20.0], [0.0, -35.0]]
    floor = Rect([500, 600], [1000, 50], restitution=0.3, fric_coeff=0.5)
    a = Hull([430, 500], [sc * torch.tensor(v, dtype=torch.double) for v in pent], vel=[1.5, 20, 0], restitution=0.3, fric_coeff=0.5)
    b = Hull([520, 520], [torch.tensor(v, dtype=torch.double) for v in tri], vel=[0, -30, 0], restitution=0.3, fric_coeff=0.5)
    assert np.abs(np.array([float(a.ang_inertia), float(b.ang_inertia)]) - g["inertia"]).max() < 1e-9
    for x in (a, b):
        x.add_force(Gravity(g=100))
    w = World([floor, a, b], [TotalConstraint(floor)], dt=1.0 / 30)
    for k in range(50):
        w.step()
        want = [tuple(r) for r in g["pairs"][k] if r[0] >= 0]
        assert [(c[1], c[2]) for c in w.contacts] == want, (k, want)
        p = torch.cat([x.p for x in (floor, a, b)]).detach().cpu().numpy()
        assert np.abs(p - g["traj_p"][k]).max() < 1e-7 * np.abs(g["traj_p"][k]).max(), k
    assert len(w.trajectory) == int(g["n_substeps"])
    loss = (a.p ** 2).sum() + (b.p ** 2).sum()
    gs, = torch.autograd.grad(loss, [sc])
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-8 * abs(float(g["loss"]))
    assert abs(float(gs) - float(g["g_scale"])) < 1e-5 * abs(float(g["g_scale"])), (float(gs), float(g["g_scale"]))


def test_malformed_pairs_are_flagged_not_read_out_of_bounds():
    """The public C ABI takes `kind` and `nv` from device memory: a vertex count beyond the table or an unknown kind must not
    index past the per-lane vertex array -- such a pair comes back with count = -1 and no contact, its neighbours untouched."""
    import torch
    from diffsdfsim_amd.physics2d.contacts import contacts2d
    dev = torch.device("cuda:0")
    P, maxv = 4, 8
    pos = torch.zeros(2, P, 2, dtype=torch.float64, device=dev); pos[1, :, 0] = 1.5
    rad = torch.ones(2, P, dtype=torch.float64, device=dev)
    verts = torch.zeros(2, P, maxv, 2, dtype=torch.float64, device=dev)
    kind = torch.zeros(2, P, dtype=torch.int32, device=dev)
    nv = torch.zeros(2, P, dtype=torch.int32, device=dev)
    sat = torch.zeros(2, P, dtype=torch.int32, device=dev)
    kind[0, 1] = 1; nv[0, 1] = 1000            # polygon with an absurd vertex count
    kind[1, 2] = 7                             # unknown kind
    out, count, _ = contacts2d(pos, rad, verts, kind, nv, sat, 0.1)
    c = count.cpu().numpy()
    assert c[0] == 1 and c[3] == 1 and c[1] == -1 and c[2] == -1, c
    assert float(out[1].abs().max()) == 0.0 and float(out[2].abs().max()) == 0.0 and float(out[0].abs().max()) > 0.0
