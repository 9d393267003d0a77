"""GPU: the reference-shaped Python surface (World3D / bodies / constraints / forces / autograd) end to end."""
import numpy as np
import pytest
import torch

import rollout_helpers as R

pytestmark = pytest.mark.gpu

# the goldens were recorded with the reference's analytic meshes / inertias (its defaults are the level-set ones)
CUSTOM = dict(custom_mesh=True, custom_inertia=True)


def build_sphere_world(g, toc):
    from diffsdfsim_amd.physics3d import Gravity3D, SDFBox, SDFSphere, TotalConstraint3D, World3D
    floor = SDFBox([0, -0.5, 0], [4.0, 1.0, 4.0], restitution=float(g["restitution"][0]), fric_coeff=float(g["fric"][0]), **CUSTOM)
    rad = torch.tensor(float(g["param_0"]), dtype=torch.float64, requires_grad=True)
    ball = SDFSphere(g["pose0"][1, 4:].tolist(), rad, vel=g["vel0"][1].tolist(), restitution=float(g["restitution"][1]),
                     fric_coeff=float(g["fric"][1]), **CUSTOM)
    ball.add_force(Gravity3D())
    w = World3D([floor, ball], [TotalConstraint3D(floor)], time_of_contact_diff=toc)
    return w, floor, ball, rad


@pytest.mark.parametrize("name,toc", [("rollout_sphere_notoc", False), ("rollout_sphere", True)])
def test_world3d_rollout_and_gradient_match_reference(name, toc):
    """Same scene as the golden built through the public API (with and without the time-of-contact differential,
    World.H); the floor mesh comes from diffsdfsim_amd.meshes (1 ulp from the reference's torch.linspace), so
    trajectories agree to ~1e-9, not 1e-12."""
    g = R.load_rollout(name)
    w, floor, ball, rad = build_sphere_world(g, toc=toc)
    for _ in range(24):
        w.step(fixed_dt=True)
    k = len(g["traj_t"]) - 1
    assert abs(w.t - float(g["t_final"])) < 1e-12
    assert np.abs(ball.p.detach().cpu().numpy() - g["traj_p"][k][1]).max() < 1e-7
    assert np.abs(ball.v.detach().cpu().numpy() - g["traj_v"][k][1]).max() < 1e-7
    loss = sum((b.pos ** 2).sum() for b in (floor, ball))
    assert abs(float(loss) - float(g["loss"])) < 1e-7
    loss.backward()
    assert abs(float(rad.grad) - float(g["grad_0"])) < 1e-5 * abs(float(g["grad_0"])) + 1e-9, (rad.grad, g["grad_0"])


def test_world3d_trajectory_entries_are_the_references():
    """`world.trajectory` as lcp_physics/physics/world.py:373-379 builds it: one entry per ACCEPTED sub-step, appended before
    `self.t += dt`, i.e. stamped with the time at the START of the sub-step and holding the state AFTER it (the experiments'
    `trajectory_loss`, optim_sphere.py:114-160, pairs entries by these times and divides by their number).  Sphere drop with
    time-of-contact events: 245 entries for 100 steps; times, poses and velocities against the reference's list, and the
    `undo_step` quirk that the undone step's first entry survives (`while self.trajectory[-1][0] > self.t`, world.py:114-116)."""
    g = R.load_rollout("rollout_sphere_long")
    w, floor, ball, rad = build_sphere_world(g, toc=True)
    for _ in range(100):
        w.step(fixed_dt=True)
    assert len(w.trajectory) == len(g["traj_t"])
    assert len(g["traj_t"]) > 100, "the scene was meant to halve dt"
    for e, t, p, v in zip(w.trajectory, g["traj_t"], g["traj_p"], g["traj_v"]):
        assert abs(float(e[0]) - float(t)) < 1e-12, (float(e[0]), float(t))
        assert np.abs(e[1].detach().cpu().numpy().reshape(2, 7) - p).max() < 1e-7 and np.abs(e[2].detach().cpu().numpy().reshape(2, 6) - v).max() < 1e-7
    n = len(w.trajectory)
    t_before = w.t
    w.step(fixed_dt=True)
    added = len(w.trajectory) - n
    w.undo_step()
    assert w.t == t_before and len(w.trajectory) == n + 1 and added >= 1      # the entry stamped t_before stays, as in the reference


def test_world3d_step_without_fixed_dt_and_run_world():
    from diffsdfsim_amd.physics3d import run_world
    g = R.load_rollout("rollout_sphere_notoc")
    w, floor, ball, rad = build_sphere_world(g, toc=True)
    run_world(w, fixed_dt=False, run_time=0.2, print_time=False)
    assert w.t >= 0.2 and len(w.trajectory) >= 6
    assert isinstance(w.contacts, list)


def test_batchworld3d_parameter_gradients_flow():
    from diffsdfsim_amd import scenes
    from diffsdfsim_amd.physics3d import BatchWorld3D
    spec = scenes.sphere_drop(8, seed=3, floor_dims=(4.0, 1.0, 4.0))
    prm = torch.tensor(spec["shape_prm"], dtype=torch.float64, requires_grad=True)
    w = BatchWorld3D(spec, params=dict(shape_prm=prm), time_of_contact_diff=False, max_substeps=256)
    for _ in range(30):
        w.step()
    (w.pose[:, :, 4:] ** 2).sum().backward()
    gr = prm.grad[:, 1, 0]
    assert torch.isfinite(gr).all() and (gr != 0).any()


def test_batchworld3d_run_is_one_node_with_the_gradients_of_the_step_loop():
    """BatchWorld3D.run(n): n outer steps per scene in one autograd node with the scenes free-running (DssWorld.steps_left).  Final
    state and d sum|pos|^2 / d radius are those of n calls of step() bit for bit; with record_substeps the entries of every
    accepted sub-step come back with their graph (a loss on an intermediate entry gives the gradient of the step loop too)."""
    from diffsdfsim_amd import scenes
    from diffsdfsim_amd.physics3d import BatchWorld3D
    out = []
    for free in (False, True):
        spec = scenes.sphere_drop(16, seed=3, floor_dims=(4.0, 1.0, 4.0))
        prm = torch.tensor(spec["shape_prm"], dtype=torch.float64, requires_grad=True)
        w = BatchWorld3D(spec, params=dict(shape_prm=prm), time_of_contact_diff=True, max_substeps=400)
        w.record_substeps = True
        if free:
            w.run(40)
            mid = w.substeps["pose"][1]                       # the state after every scene's 2nd accepted sub-step
        else:
            mid = None
            for i in range(40):
                w.step(keep_undo=False)
                if i == 1:
                    mid = w.pose                              # (no dt halving in the first two steps: sub-step 2 = step 2)
        loss = (w.pose[:, :, 4:] ** 2).sum() + 0.5 * (mid[:, :, 4:] ** 2).sum()
        loss.backward()
        out.append((w.pose.detach().cpu().numpy().copy(), w.engine.get("nsub").copy(), prm.grad.detach().cpu().numpy().copy(), float(loss)))
    (p0, n0, g0, l0), (p1, n1, g1, l1) = out
    assert int(n0.max()) > 40                                 # somebody bounced
    assert np.array_equal(p0, p1) and np.array_equal(n0, n1) and l0 == l1
    assert np.abs(g0).max() > 0 and np.abs(g0 - g1).max() <= 1e-12 * np.abs(g0).max()


def test_engine_plugin_solve_dynamics_matches_the_step():
    """B2: HipPdipmEngine.solve_dynamics(world, dt) (engines.py:31-83) returns the velocities the next accepted
    sub-step integrates with, without advancing the world."""
    from diffsdfsim_amd.physics3d import HipPdipmEngine
    g = R.load_rollout("rollout_sphere_notoc")
    w, floor, ball, rad = build_sphere_world(g, toc=False)
    for _ in range(9):
        w.step(fixed_dt=True)
    p_before = w.pose.clone()
    v_new = w.engine_plugin.solve_dynamics(w, w.dt)
    assert isinstance(w.engine_plugin, HipPdipmEngine)
    assert torch.equal(w.pose, p_before)
    n0 = int(w.engine.get("nsub")[0])
    w.step(fixed_dt=True)
    if int(w.engine.get("nsub")[0]) == n0 + 1:      # accepted at the first attempt: same dt, same solve
        assert torch.allclose(v_new.reshape(2, 6), w.vel[0], rtol=0, atol=1e-14)


def test_undo_step_set_p_set_v():
    """World surface used by the experiments (lcp_physics/physics/world.py:106-116, 381-391): undo_step returns to
    the start of the last step (time, poses, velocities, contacts, trajectory), after which the same step repeats
    bit for bit; set_v / set_p replace the state of all bodies."""
    g = R.load_rollout("rollout_sphere_notoc")
    w, floor, ball, rad = build_sphere_world(g, toc=True)
    for _ in range(10):
        w.step(fixed_dt=True)
    t0, p0, v0, nc0, ntraj = w.t, w.pose.clone(), w.v.clone(), len(w.contacts), len(w.trajectory)
    w.step(fixed_dt=True)
    p1, v1, t1 = w.pose.clone(), w.v.clone(), w.t
    w.undo_step()
    assert w.t == t0 and torch.equal(w.pose, p0) and torch.equal(w.v, v0)
    # the trajectory keeps the undone step's FIRST entry, as in the reference: entries are stamped with the start time of their
    # sub-step and undo_step pops `while self.trajectory[-1][0] > self.t` (world.py:114-116) -- that entry's stamp equals self.t
    assert len(w.contacts) == nc0 and len(w.trajectory) == ntraj + 1 and torch.equal(ball.p, p0[0, 1])
    w.step(fixed_dt=True)
    assert w.t == t1 and torch.equal(w.pose, p1) and torch.equal(w.v, v1)
    nv = w.v.clone(); nv[9] = 0.25                      # give the ball a push along x
    w.set_v(nv)
    npose = w.pose.reshape(-1).clone(); npose[7 + 5] += 0.1
    w.set_p(npose)
    assert float(ball.v[3]) == 0.25 and abs(float(ball.pos[1]) - float(p1[0, 1, 5]) - 0.1) < 1e-15
    w.step(fixed_dt=True)
    assert torch.isfinite(w.pose).all() and w.observations == []


def test_contact_free_constrained_body_matches_reference():
    """Config-5 shape: one body, X/Y/ZConstraint, a torque for t < 0.3, no contacts: the engine's
    linear-solve branch (engines.py:40-54), time-dependent ExternalForce3D, gradient through the analytic inertia."""
    import os
    from diffsdfsim_amd.physics3d import ExternalForce3D, SDFBox, World3D, XConstraint, YConstraint, ZConstraint
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "config5_spin.npz"))
    dims = torch.tensor(g["dims"], dtype=torch.float64, requires_grad=True)
    body = SDFBox([0, 0, 0], dims, mass=float(g["mass"]), custom_mesh=True, custom_inertia=True)
    tq = torch.tensor(np.concatenate([g["dir"], np.zeros(3)]))
    body.add_force(ExternalForce3D(lambda t: tq if t < float(g["t_off"]) else ExternalForce3D.ZEROS, multiplier=float(g["mag"])))
    w = World3D([body], [XConstraint(body), YConstraint(body), ZConstraint(body)])
    for k in range(int(g["nsteps"])):
        w.step(fixed_dt=True)
        assert abs(w.t - (g["traj_t"][k] + float(g["dt"]))) < 1e-12    # the reference stamps an entry with its start time
        assert np.abs(body.p.detach().cpu().numpy() - g["traj_p"][k]).max() < 1e-10
        assert np.abs(body.v.detach().cpu().numpy() - g["traj_v"][k]).max() < 1e-10
    loss = (body.v[:3] ** 2).sum()
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-10
    assert np.abs(dims.grad.numpy() - g["grad_dims"]).max() < 1e-5 * np.abs(g["grad_dims"]).max()


def test_body_set_p_and_mass_matrix():
    """Body surface of the reference (bodies.py:431-435, 498-511): `M` = blockdiag(R I R^T, m 1); a pose replaced with
    `body.set_p` is what the world steps from."""
    g = R.load_rollout("rollout_sphere_notoc")
    w, floor, ball, rad = build_sphere_world(g, toc=True)
    M = ball.M.detach()
    assert torch.allclose(M[3:, 3:], torch.eye(3, dtype=M.dtype, device=M.device) * float(ball.mass))
    assert torch.allclose(M[:3, :3], ball.ang_inertia.to(M), atol=1e-15)          # identity rotation, isotropic inertia
    w.step(fixed_dt=True)
    p = ball.p.detach().clone(); p[5] += 0.25
    ball.set_p(p)
    y0 = float(ball.pos[1])
    w.step(fixed_dt=True)
    assert abs(float(w.pose[0, 1, 5]) - y0) < 0.05 and float(w.pose[0, 1, 5]) > float(g["traj_p"][1][1][5]) + 0.2


def test_world3d_three_bodies_gradients_wrt_box_dims_and_sphere_radius():
    """`rollout_sphere_on_box` (floor, a box on it, a sphere dropped on the box) through the class API: the gradient of
    sum|pos_T|^2 w.r.t. the box's dims and the sphere's radius flows through shape parameters AND the analytic inertias
    (torch), against the reference's autograd."""
    from diffsdfsim_amd.physics3d import Gravity3D, SDFBox, SDFSphere, TotalConstraint3D, World3D
    g = R.load_rollout("rollout_sphere_on_box")
    floor = SDFBox([0, -0.5, 0], [4.0, 1.0, 4.0], restitution=0.3, fric_coeff=0.4, **CUSTOM)
    d = torch.tensor(g["param_0"], dtype=torch.float64, requires_grad=True)
    r = torch.tensor(float(g["param_1"]), dtype=torch.float64, requires_grad=True)
    box = SDFBox(g["pose0"][1, 4:].tolist(), d, vel=g["vel0"][1].tolist(), restitution=0.2, fric_coeff=0.4, **CUSTOM)
    ball = SDFSphere(g["pose0"][2, 4:].tolist(), r, vel=g["vel0"][2].tolist(), restitution=0.2, fric_coeff=0.4, **CUSTOM)
    for b in (box, ball):
        b.add_force(Gravity3D())
    w = World3D([floor, box, ball], [TotalConstraint3D(floor)], time_of_contact_diff=True)
    for _ in range(12):
        w.step(fixed_dt=True)
    k = len(g["traj_t"]) - 1
    for i, b in enumerate((floor, box, ball)):
        assert np.abs(b.p.detach().cpu().numpy() - g["traj_p"][k][i]).max() < 1e-7
    loss = sum((b.pos ** 2).sum() for b in (floor, box, ball))
    loss.backward()
    assert np.abs(d.grad.numpy() - g["grad_0"]).max() < 1e-5 * np.abs(g["grad_0"]).max(), (d.grad, g["grad_0"])
    assert abs(float(r.grad) - float(g["grad_1"])) < 1e-5 * abs(float(g["grad_1"])), (r.grad, g["grad_1"])


def test_loss_on_an_intermediate_step_with_the_world_stepped_further():
    """d |pos_k|^2 / d rad for k < T, the world having been stepped on to T before backward(): the reverse sweep starts at
    the tape slot of step k, not at the engine's newest one.  Must equal the gradient of a world stepped to k only."""
    g = R.load_rollout("rollout_sphere")

    def grad_at(k, T):
        w, floor, ball, rad = build_sphere_world(g, toc=True)
        loss = None
        for i in range(T):
            w.step(fixed_dt=True)
            if i + 1 == k:
                loss = (ball.pos ** 2).sum()
        loss.backward()
        return float(rad.grad)

    a, b = grad_at(20, 20), grad_at(20, 24)
    assert a != 0.0 and a == b, (a, b)


def test_world3d_can_be_put_on_the_full_kernel_variants():
    """`full_kernels=True` (an extra keyword: the exact hull of big contact clusters, every primitive) gives the same step as
    the lean variants on a scene both can run."""
    import torch
    from diffsdfsim_amd.physics3d import Gravity3D, SDFBox, TotalConstraint3D, World3D
    out = []
    for full in (None, True):
        floor = SDFBox([0, -0.5, 0], [4.0, 1.0, 4.0], custom_mesh=True, custom_inertia=True)
        b = SDFBox([0.0, 0.2505, 0.0], [0.5, 0.5, 0.4], vel=[0.2, 0.1, 0, 0.3, 0, 0], custom_mesh=True, custom_inertia=True)
        b.add_force(Gravity3D())
        w = World3D([floor, b], [TotalConstraint3D(floor)], full_kernels=full)
        assert int(w.engine.W.shape_rare) == int(bool(full))
        for _ in range(5):
            w.step(fixed_dt=True)
        out.append(b.p.detach().cpu().clone())
    assert torch.equal(out[0], out[1])
