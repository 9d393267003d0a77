"""GPU: a NEURAL SDF body inside the batched stepper (BASELINE configs[3], demos/demo_meshsdf.py) against rollouts recorded
from the reference's own ``SDF3D.query_sdfs`` / ``FWContactHandler`` / ``World3D`` with a seeded stand-in network
(oracle/gen/gen_igr_golden.py; the trained IGR weights are not available offline, so the network itself stays
parity-unpinned -- the STEPPER around it is pinned here).

north_star tolerance: contact-pair indices exact, positions / velocities / gradients 1e-5 relative."""
import numpy as np
import pytest

import igr_helpers as H
import rollout_helpers as R

pytestmark = pytest.mark.gpu


def run_like_run_world(E, g):
    """`run_world`'s loop (physics3d/world.py:113-205): world.step() -- one step_dt each -- until t >= run_time."""
    n = 0
    while float(E.get("t")[0]) < float(g["run_time"]):
        E.step_once()
        n += 1
        assert n < 500
    return n


@pytest.mark.parametrize("name", ["rollout_igr_small", "rollout_igr_demo"])
def test_neural_body_rollout_matches_reference(name):
    from diffsdfsim_amd.engine import BatchEngine
    g = R.load_rollout(name)
    E = BatchEngine(H.spec_from_golden(g, 2), **H.engine_kwargs(g, max_sub=128))
    assert int(E.get("overflow").max()) == 0
    R.check_contacts(E, 0, g["init_body"], g["init_geom"], len(g["init_body"]))
    run_like_run_world(E, g)
    assert int(E.get("overflow").max()) == 0
    nsub = E.get("nsub")
    assert (nsub == len(g["traj_t"])).all(), (nsub, len(g["traj_t"]))
    k = len(g["traj_t"]) - 1
    pose, vel = E.get("pose"), E.get("vel")
    tp, tnc, tb = E.get("tp_pose"), E.get("tp_nc"), E.get("tp_body")
    # every step of the way: start poses of sub-step j+1 = the reference's poses after sub-step j; same contact pairs
    for j in range(1, k + 1):
        assert np.abs(tp[j, 0] - g["traj_p"][j - 1]).max() < 1e-7, j
        n = int(g["traj_nc"][j - 1])
        assert int(tnc[j, 0]) == n, (j, int(tnc[j, 0]), n)
        assert [tuple(r) for r in tb[j, 0][:, :n].T] == [tuple(r) for r in g["traj_body"][j - 1][:n]], j
    scale = max(1.0, np.abs(g["traj_p"][k]).max())
    assert np.abs(pose[0] - g["traj_p"][k]).max() < 1e-7 * scale and np.abs(vel[0] - g["traj_v"][k]).max() < 1e-6
    assert (pose == pose[:1]).all() and (vel == vel[:1]).all(), "replicated scenes diverged"
    for s in (0, 1):
        R.check_contacts(E, s, g["traj_body"][k], g["traj_geom"][k], int(g["traj_nc"][k]))


def build_world(g, latent):
    """The golden's scene through the reference-shaped public surface (demos/demo_meshsdf.py:121-142 for rollout_igr_demo)."""
    from diffsdfsim_amd.physics3d import Gravity3D, SDF3D, SDFBox, SDFCylinder, TotalConstraint3D, World3D
    from diffsdfsim_amd.physics3d.utils import decode_igr
    net = H.torch_network(g)
    bodies, joints = [], []
    for i, kind in enumerate(g["kind"]):
        p0 = g["pose0"][i]
        kw = dict(fric_coeff=float(g["fric"][i]), restitution=float(g["restitution"][i]))
        if kind == 0:
            b = SDFBox(p0.tolist(), g["shape_prm"][i].tolist(), custom_mesh=bool(g["custom_mesh"][i]), custom_inertia=bool(g["custom_mesh"][i]), **kw)
        elif kind == 2:
            b = SDFCylinder(p0.tolist(), float(g["shape_prm"][i][0]), float(g["shape_prm"][i][1]), **kw)
        else:
            b = SDF3D(pos=p0.tolist(), scale=float(g["igr_scale"]), sdf_func=decode_igr(net), params=[latent], vel=g["vel0"][i].tolist(), **kw)
            b.add_force(Gravity3D())
            obj = b
        bodies.append(b)
        if i in g["fixed"]:
            joints.append(TotalConstraint3D(b))
    nc = g["no_contact"]
    for i in range(len(bodies)):
        for j in range(i + 1, len(bodies)):
            if nc[i, j]:
                bodies[i].add_no_contact(bodies[j])
    return World3D(bodies, joints), obj


@pytest.mark.parametrize("name", ["rollout_igr_small", "rollout_igr_demo"])
def test_latent_gradient_matches_reference_autograd(name):
    """d loss / d latent of the demo's loss (demos/demo_meshsdf.py:89) through the whole rollout: through the network's value at
    the contact points (the stepper's reverse sweep, matrix cores), through the level-set mesh (vertex adjoint -> MeshSDF
    backward) and through the inertia integrated over that mesh.  The mesh here is the device's own (the network on the
    matrix cores, not torch's CPU matmul), so poses agree to ~1e-9 rather than bit for bit."""
    import torch
    from diffsdfsim_amd.physics3d import run_world
    g = R.load_rollout(name)
    latent = torch.tensor(g["latent"], dtype=torch.float64, requires_grad=True)
    w, obj = build_world(g, latent)
    assert tuple(g["meshsize_%d" % int(g["igr_body"])]) == (len(obj.verts_np), len(obj.faces_np))
    run_world(w, run_time=float(g["run_time"]), print_time=False)
    assert len(w.trajectory) == len(g["traj_t"])
    k = int(g["igr_body"])
    assert np.abs(obj.p.detach().cpu().numpy() - g["traj_p"][-1][k]).max() < 1e-6
    loss = (obj.pos - torch.tensor(g["target"], device=obj.pos.device)).norm() ** 2 + 0.05 * latent.norm() ** 2
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-6
    loss.backward()
    got, want = latent.grad.cpu().numpy(), g["grad_latent"]
    assert np.abs(got - want).max() < 1e-5 * np.abs(want).max(), (got, want)


def test_system_identification_scene_gradients_match_reference_autograd():
    """The scene of experiments/system_identification/optim_sysid.py:104-131 -- a neural-SDF body set down on the floor and
    pushed along it, strict_no_penetration=False -- through the public surface: eight outer steps (14 sub-steps with dt
    halving), the same trajectory as the reference, and d sum_t |pos_t - target_t|^2 / d (push, mass, friction coefficient)
    equal to the reference's autograd values: the mass through inertia, gravity and the LCP's mass blocks, the friction
    coefficient (one tensor for both bodies) through the cone rows, the push through the external force."""
    import torch
    from diffsdfsim_amd.physics3d import ExternalForce3D, Gravity3D, SDF3D, SDFBox, TotalConstraint3D, World3D
    from diffsdfsim_amd.physics3d.utils import decode_igr, get_tensor
    g = R.load_rollout("rollout_igr_push")
    net = H.torch_network(g)
    latent = torch.tensor(g["latent"], dtype=torch.float64)
    force = torch.tensor(g["force"], dtype=torch.float64, requires_grad=True)
    mass = torch.tensor([float(g["mass"])], dtype=torch.float64, requires_grad=True)
    fric = torch.tensor([float(g["fric"])], dtype=torch.float64, requires_grad=True)

    def force_func(t):
        fv = get_tensor([0, 0, 0, 0, 0, 0])
        idx = torch.tensor([3, 5])
        return fv.index_put((idx,), force.to(fv))
    floor = SDFBox([0, -.5, 0], [20, 1, 20], fric_coeff=fric, restitution=0.0, custom_mesh=True, custom_inertia=True)
    obj = SDF3D([0, 0, 0], scale=1, sdf_func=decode_igr(net), params=[latent], mass=mass, fric_coeff=fric, restitution=0.0)
    assert (len(obj.verts_np), len(obj.faces_np)) == tuple(g["meshsize_1"])
    obj.set_p(torch.tensor(g["pose0"][1], dtype=torch.float64))
    obj.add_force(Gravity3D())
    obj.add_force(ExternalForce3D(force_func))
    w = World3D([floor, obj], [TotalConstraint3D(floor)], time_of_contact_diff=True, strict_no_penetration=False, fric_dirs=8)
    loss = 0.0
    for k in range(int(g["nsteps"])):
        w.step(fixed_dt=True)
        loss = loss + ((torch.tensor(g["target"][k], device=obj.pos.device) - obj.pos) ** 2).sum()
    assert len(w.trajectory) == len(g["traj_t"])
    assert np.abs(obj.p.detach().cpu().numpy() - g["traj_p"][-1][1]).max() < 1e-6
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-7
    loss.backward()
    for name, t in (("grad_force", force), ("grad_mass", mass), ("grad_fric", fric)):
        got, want = t.grad.cpu().numpy(), g[name]
        assert np.abs(got - want).max() < 1e-5 * np.abs(want).max(), (name, got, want)


def test_neural_body_scenes_free_running_reproduce_lock_step():
    """bench.py --config 4 steps with BatchEngine.run(K) (DssWorld.steps_left): scenes with a neural SDF body go through their
    outer steps independently while their network queries still share the launches of a round.  Four copies of the
    `rollout_igr_small` scene with the neural body started at different heights, 12 outer steps both ways: poses, velocities,
    times, sub-step counts, contacts and the tape are identical bit for bit (a query's result depends on the point alone)."""
    from diffsdfsim_amd.engine import BatchEngine
    g = R.load_rollout("rollout_igr_small")
    out = []
    for free in (False, True):
        spec = H.spec_from_golden(g, 4)
        mv = [b for b in range(spec["pose"].shape[1]) if b not in g["fixed"]][0]        # the neural body
        f = spec["fext"][0, mv, 3:]
        up = -f / np.linalg.norm(f)
        for s, dz in enumerate((0.0, 0.011, 0.023, 0.037)):
            spec["pose"][s, mv, 4:] += dz * up
        E = BatchEngine(spec, **H.engine_kwargs(g, max_sub=256))
        rounds = E.run(12) if free else sum(E.step() for _ in range(12))
        assert int(E.get("overflow").max()) == 0
        out.append((rounds, {k: E.get(k).copy() for k in ("pose", "vel", "t", "nsub", "nc", "c_geom", "tp_pose", "tp_vel", "tp_dt", "tp_nc", "tp_lam")}))
    (r0, a), (r1, b) = out
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    print("attempt rounds: lock-step %d, free-running %d; sub-steps per scene %s" % (r0, r1, a["nsub"]))
    assert len(set(a["nsub"])) > 1, "the scenes were meant to differ"
    assert r1 <= r0
