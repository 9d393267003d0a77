"""GPU: batched experiment drivers (diffsdfsim_amd/experiments.py): the loops of
`experiments/trajectory_fitting/optim_sphere.py` and `experiments/inertia_fitting/optim_shapespace.py` for many scenes at once."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_bounce_scene_undo_and_detach_are_per_scene():
    """`run_world_fixed_dt(..., detach_2nd_bounce)` (optim_sphere.py:163-177) on a batch whose scenes hit floor and wall in
    different steps: every scene ends at the same time having taken its own number of outer steps (an undone step is redone),
    and a scene run alone goes through exactly the same states."""
    from diffsdfsim_amd import experiments as X
    rad = torch.tensor([0.5, 0.9, 1.4, 0.7], dtype=torch.float64)
    w = X.bounce_world(rad, run_time=1.0)
    tr = X.run_world_fixed_dt(w, 1.0, detach_2nd_bounce=True)
    assert tr["valid"].shape[1] == 4 and bool((torch.as_tensor(w.t) >= 1.0 - 1e-9).all())
    n_valid = tr["valid"].sum(dim=0).cpu().numpy()
    # one entry per accepted sub-step: at least the 30 outer steps of 1/30 s, more where dt was halved around a contact
    assert (n_valid >= 30).all() and n_valid.max() > 31, n_valid
    assert tr["valid"].shape[0] > n_valid.min(), "some step was meant to be undone and redone (second contact step in a row)"
    one = X.run_world_fixed_dt(X.bounce_world(rad[2:3], run_time=1.0), 1.0, detach_2nd_bounce=True)
    a = tr["pose"][:, 2][tr["valid"][:, 2]]
    b = one["pose"][:, 0][one["valid"][:, 0]]
    assert a.shape == b.shape and torch.equal(a, b)


def test_radius_fitting_on_the_bounce_scene():
    """The gradient-descent loop of optim_sphere.py:210-270 for eight (target, start) pairs at once: losses fall, radii move
    towards their targets, the chamfer distance between the fitted and the target sphere shrinks with them."""
    from diffsdfsim_amd import experiments as X, meshes
    r = np.random.default_rng(5)
    B = 8
    target = 0.5 + 0.8 * r.random(B)
    start = target + np.where(r.random(B) < 0.5, -0.15, 0.15)
    res = X.fit_sphere_radius(target, start, run_time=1.0, max_iter=12, lr=0.1)
    h = res["history"]
    e0, e1 = np.abs(start - target), np.abs(res["radius"] - target)
    assert np.isfinite(h[-1]["grad"]).all()
    assert h[-1]["loss"].mean() < 0.5 * h[0]["loss"].mean(), (h[0]["loss"].mean(), h[-1]["loss"].mean())
    assert e1.mean() < 0.7 * e0.mean(), (e0.mean(), e1.mean())
    uv, _ = meshes.icosphere(3)
    uv = torch.as_tensor(uv)
    d0 = float(X.chamfer(uv * start[0], uv * target[0])); d1 = float(X.chamfer(uv * res["radius"][0], uv * target[0]))
    assert (d1 < d0) == (e1[0] < e0[0])
    print(X.radius_error_table({"gravity, toc (8 scenes, 12 iterations)": res}))


def test_trajectory_loss_pairs_by_nearest_time():
    from diffsdfsim_amd import experiments as X
    t = torch.tensor([[0.1], [0.2], [0.2], [0.3]], dtype=torch.float64)
    mk = lambda pos, valid: dict(t=t, pose=torch.cat([torch.zeros(4, 1, 1, 4, dtype=torch.float64), pos], dim=3), vel=None,
                                 valid=torch.tensor(valid)[:, None])
    tgt = mk(torch.tensor([1.0, 2.0, 9.0, 3.0], dtype=torch.float64).reshape(4, 1, 1, 1).expand(4, 1, 1, 3), [True, True, False, True])
    src = mk(torch.tensor([1.5, 9.0, 2.5, 3.5], dtype=torch.float64).reshape(4, 1, 1, 1).expand(4, 1, 1, 3), [True, False, True, True])
    # valid entries 0, 2, 3 at t = .1, .2, .3 against target values 1, 2, 3: three coordinates each 0.5 off
    assert abs(float(X.trajectory_loss(src, tgt)[0]) - 0.75) < 1e-12


def test_inertia_fitting_gradient_reaches_the_latent_code():
    """optim_shapespace.py:136-250 for two scenes, low mesh resolution: the loss (final angular velocity under the torque)
    falls, the gradient arrives at the latent code through mesh, volume integrals and every step's linear solve."""
    from diffsdfsim_amd import experiments as X, igr, scenes
    packed = igr.pack_weights(*scenes.geometric_init_weights(0, 0.5))
    tgt = np.array([[0.05, -0.08], [-0.06, 0.03]]); st = tgt + np.array([[0.06, 0.05], [0.05, -0.06]])
    dirs = np.array([[1.0, 0.0, 0.0], [0.0, 0.6, 0.8]])
    res = X.fit_inertia_latent(tgt, st, dirs, packed, run_time=0.5, max_iter=4, lr=2e-2, res=48)
    h = res["history"]
    assert np.isfinite(h[0]["grad"]).all() and np.abs(h[0]["grad"]).max() > 0
    assert h[-1]["loss"].sum() < h[0]["loss"].sum(), ([x["loss"] for x in h])


def test_trajectory_fitting_in_shape_space_reaches_the_latent_code():
    """trajectory_fitting/optim_shapespace.py for two scenes at low mesh resolution: a neural body thrown at the wall, the
    trajectory loss against the target shape's flight; the gradient arrives at the latent code (through the contact's SDF
    queries, the level-set mesh and the inertia) and a few small plain-gradient steps lower the loss."""
    from diffsdfsim_amd import experiments as X, igr, scenes
    packed = igr.pack_weights(*scenes.geometric_init_weights(0, 0.5))
    tgt = np.array([[0.05, -0.08], [-0.06, 0.03]]); st = tgt + np.array([[0.08, 0.06], [0.06, -0.08]])
    res = X.fit_trajectory_latent(tgt, st, packed, run_time=1.2, max_iter=4, lr=5e-3, optimizer="GD", latent_reg=0.0, res=48)
    h = res["history"]
    assert np.isfinite(h[0]["grad"]).all() and (np.abs(h[0]["grad"]).max(axis=1) > 0).all()
    assert (h[0]["loss"] > 0).all()
    assert h[-1]["loss"].sum() < h[0]["loss"].sum(), [x["loss"] for x in h]


@pytest.mark.parametrize("kind", ["box", "sphere", "cylinder"])
def test_inertia_fitting_of_primitive_dimensions(kind):
    """optim_primitives.py:160-240 for two scenes: the dimensions of an SDFBox / SDFSphere / SDFCylinder (marching-cubes
    mesh and mesh inertia, the experiment's defaults) move towards the target's under Adam, the loss falls, and the first
    gradient has the sign the physics dictates: a body bigger than its target spins up less, so the loss grows with its size."""
    from diffsdfsim_amd import experiments as X
    n = X.PRIMITIVES[kind]
    tgt = np.array([[1.0, 1.4, 0.8], [1.2, 0.9, 1.1]])[:, :n]
    st = tgt * np.array([[1.25], [0.8]])
    dirs = np.array([[1.0, 0.0, 0.0], [0.0, 0.6, 0.8]])
    res = X.fit_inertia_primitive(kind, tgt, st, dirs, run_time=0.5, max_iter=4, lr=2e-2)
    h = res["history"]
    assert np.isfinite(h[0]["grad"]).all()
    # scene 0 started too big in every dimension, scene 1 too small (a dimension along the torque axis does not enter: its
    # derivative is discretisation noise of the level-set mesh, so the sign is asked of the sum)
    assert h[0]["grad"][0].sum() > 0 and h[0]["grad"][0].min() > -0.1 * h[0]["grad"][0].max()
    assert h[0]["grad"][1].sum() < 0
    assert h[-1]["loss"].sum() < h[0]["loss"].sum(), [x["loss"] for x in h]
    assert np.abs(res["dims"] - tgt).mean() < np.abs(st - tgt).mean()
    assert res["dims"].min() >= 0.5 and res["dims"].max() <= 2.0


@pytest.mark.parametrize("goal", ["mass", "force", "friction"])
def test_system_identification_of_a_pushed_neural_body(goal):
    """experiments/system_identification/optim_sysid.py:104-300 for three scenes at once (short horizon, coarse mesh): a neural
    SDF body pushed along the floor; the gradient of sum_t |pos_t - pos_t*|^2 reaches the goal (mass through inertia, gravity
    and the LCP's mass blocks; the push through the external force; the friction coefficient through the cone rows), the loss
    falls and the estimate moves towards the target."""
    from diffsdfsim_amd import experiments as X, igr, scenes
    packed = igr.pack_weights(*scenes.geometric_init_weights(0, 0.5))
    lat = np.array([[0.05, -0.03], [-0.04, 0.06], [0.0, 0.02]])
    target = dict(force=np.array([[3.0, 2.5], [4.0, 2.0], [2.5, 3.5]]), mass=np.array([1.0, 0.95, 1.05]), fric=np.array([0.1, 0.2, 0.15]))
    start = dict(force=target["force"] + np.array([[0.8, -0.6], [-0.7, 0.9], [0.6, 0.5]]), mass=target["mass"] + np.array([0.08, -0.07, 0.06]),
                 fric=target["fric"] + np.array([0.06, -0.08, 0.07]))
    res = X.fit_sysid(goal, lat, packed, target, start, run_time=0.4, max_iter=6, res=48)
    h = res["history"]
    assert np.isfinite(h[0]["grad"]).all() and np.abs(h[0]["grad"]).min() > 0
    assert h[-1]["loss"].sum() < h[0]["loss"].sum(), [x["loss"].sum() for x in h]
    assert h[-1]["dist"].mean() < h[0]["dist"].mean(), (h[0]["dist"], h[-1]["dist"])


def test_trajectory_loss_matches_the_reference_on_a_bounce_that_halves_dt():
    """The reference's own `make_world` / `run_world_fixed_dt` / `trajectory_loss` (optim_sphere.py:77-177), imported as they are,
    on the wall + floor bounce (target radius 0.7, start 0.9; oracle/gen/gen_trajloss_golden.py): the sphere hits the wall inside a
    step, dt is halved, and the trajectory gets extra entries around the contact which the loss counts like any other.  Held:
    the number of entries, their time stamps (start of the sub-step), the loss and d loss / d radius (1e-5), without and with
    detach_2nd_bounce (whose undone step leaves its first entry behind, as in the reference)."""
    import os
    from diffsdfsim_amd import experiments as X
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "trajectory_loss_bounce.npz"))
    run_time = float(g["run_time"])
    with torch.no_grad():
        tgt = X.run_world_fixed_dt(X.bounce_world(torch.tensor([float(g["r_target"])], dtype=torch.float64), run_time=run_time), run_time)
    tv = tgt["valid"][:, 0].cpu().numpy()
    assert int(tv.sum()) == len(g["target_t"]), (int(tv.sum()), len(g["target_t"]))
    assert np.abs(tgt["t"][:, 0].cpu().numpy()[tv] - g["target_t"]).max() < 1e-12
    assert np.abs(tgt["pose"][:, 0].cpu().numpy()[tv] - g["target_p"]).max() < 1e-7
    for tag, detach in (("plain", False), ("detach", True)):
        rad = torch.tensor([float(g["r_start"])], dtype=torch.float64, requires_grad=True)
        tr = X.run_world_fixed_dt(X.bounce_world(rad, run_time=run_time), run_time, detach_2nd_bounce=detach)
        v = tr["valid"][:, 0].cpu().numpy()
        assert int(v.sum()) == len(g[tag + "_t"]), (tag, int(v.sum()), len(g[tag + "_t"]))
        assert np.abs(tr["t"][:, 0].detach().cpu().numpy()[v] - g[tag + "_t"]).max() < 1e-12
        assert np.abs(tr["pose"][:, 0].detach().cpu().numpy()[v] - g[tag + "_p"]).max() < 1e-7
        loss = X.trajectory_loss(tr, tgt)
        loss.sum().backward()
        assert abs(float(loss[0]) - float(g[tag + "_loss"])) < 1e-7 * abs(float(g[tag + "_loss"])), (tag, float(loss[0]), float(g[tag + "_loss"]))
        assert abs(float(rad.grad[0]) - float(g[tag + "_grad"])) < 1e-5 * abs(float(g[tag + "_grad"])), (tag, float(rad.grad[0]), float(g[tag + "_grad"]))
