"""Golden rollout of BASELINE configs[3]'s scene -- ``demos/demo_meshsdf.make_world`` (demo_meshsdf.py:121-142): floor,
pinned cylinder pole, a neural-SDF body ``SDF3D(pos=[0,6,0], scale=2, sdf_func=decode_igr(net), params=[latent])``
dropped onto the pole -- through the reference's own ``SDF3D.query_sdfs`` / ``FWContactHandler`` / ``World3D``.

Run in the build container only:  python -m oracle.gen.gen_igr_golden [name ...]
The network is the stand-in of oracle/refshim/fake_igr.py (seeded geometric-init weights; the IGR repository and its
trained weights are not available offline).  Stored: the scene, the neural body's level-set mesh as the reference built it
(so both sides search the same triangles), every step of ``run_world``'s loop (``world.step()`` until t >= run_time:
t, poses, velocities, ordered contact pairs, contact geometry, and for every contact which body's normal the Laplacian
comparison of contacts.py:184-198 picked), the demo's loss (demo_meshsdf.py:89) and d loss / d latent from torch.autograd.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refshim  # noqa: E402

refshim.install()
from oracle.refshim import fake_igr  # noqa: E402
from oracle.gen.gen_rollout_golden import contacts_arrays, MAXC  # noqa: E402
import sdf_physics.physics3d.contacts as ref_contacts  # noqa: E402
from sdf_physics.physics3d.bodies import SDF3D, SDFBox, SDFCylinder  # noqa: E402
from sdf_physics.physics3d.constraints import TotalConstraint3D  # noqa: E402
from sdf_physics.physics3d.forces import Gravity3D  # noqa: E402
from sdf_physics.physics3d.utils import decode_igr, get_tensor  # noqa: E402
from sdf_physics.physics3d.world import World3D  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# which body's normal every contact of the last differentiable _compute_contacts call used (contacts.py:198)
_STABLE = []
_orig_compute = ref_contacts._compute_contacts


def _recording_compute(b1, b2, abc, contact_inds, eps=1e-3, detach_contact_b2=True):
    out = _orig_compute(b1, b2, abc, contact_inds, eps=eps, detach_contact_b2=detach_contact_b2)
    if torch.is_grad_enabled() and contact_inds.nelement() > 0:
        n, p1 = out[0].detach(), out[1]
        n2 = ref_contacts.quaternion_apply(b2.rot, b2.query_sdfs(
            ref_contacts.quaternion_apply(ref_contacts.quaternion_invert(b2.rot),
                                          (p1 + b1.pos - b2.pos)).detach())[1]).detach()
        _STABLE.append(((n - n2).norm(dim=1) < 1e-9).numpy())
    return out


ref_contacts._compute_contacts = _recording_compute


def demo_world(decoder, latent):
    """demos/demo_meshsdf.py:121-142 (restated call by call; importing the demo would pull in its render stack)."""
    fr = 0.15
    floor = SDFBox([0, -0.5, 0], [50, 1, 50], fric_coeff=fr, restitution=0)
    pole = SDFCylinder([np.pi / 2, 0, 0, 0.35, 1, 0], 0.2, 2, fric_coeff=fr)
    pole.add_no_contact(floor)
    m = SDF3D(pos=[0, 6, 0], scale=2, sdf_func=decode_igr(decoder), params=[latent], fric_coeff=fr, restitution=0)
    m.add_force(Gravity3D())
    bodies = [floor, pole, m]
    return World3D(bodies, [TotalConstraint3D(floor), TotalConstraint3D(pole)]), m, bodies


def small_world(decoder, latent, gap=0.03, vel=(0, 0, 0.3, 0.4, -1.0, 0.1)):
    """A quick variant: analytic floor, the neural body released `gap` above it, moving down and sideways."""
    fr = 0.3
    with torch.no_grad():
        v0, _f0 = SDF3D._diff_marching_cubes(decode_igr(decoder))(latent.detach())
    y0 = float(-2.0 * v0[:, 1].min()) + gap
    floor = SDFBox([0, -0.5, 0], [6.0, 1.0, 6.0], fric_coeff=fr, restitution=0.2, custom_mesh=True, custom_inertia=True)
    floor._golden_custom_mesh = True
    m = SDF3D(pos=[0, y0, 0], scale=2, sdf_func=decode_igr(decoder), params=[latent], vel=list(vel), fric_coeff=fr,
              restitution=0.2)
    m.add_force(Gravity3D())
    bodies = [floor, m]
    return World3D(bodies, [TotalConstraint3D(floor)]), m, bodies


def run(name, make, run_time, seed=0, radius=0.5, latent0=(0.05, -0.08), fixed=(0, 1), target=(0.0, 0.64, 0.0)):
    t0 = time.time()
    net, _Ws, _bs = fake_igr.seeded_net(seed, radius)
    latent = torch.tensor(latent0, dtype=torch.float64, requires_grad=True)
    w, obj, bodies = make(net, latent)
    print(name, "world built in %.1f s; meshes" % (time.time() - t0), [len(b.faces) for b in bodies])
    nb = len(bodies)
    d = dict(igr_seed=seed, igr_radius=radius, latent=np.array(latent0), igr_body=bodies.index(obj), igr_scale=float(obj.scale),
             dt=w.dt, eps=w.eps, tol=w.tol, fric_dirs=w.fric_dirs, toc_diff=1, strict_no_pen=int(w.strict_no_pen),
             fixed=np.array(fixed, np.int32), run_time=run_time, target=np.array(target))
    d["kind"] = np.array([0 if isinstance(b, SDFBox) else (2 if isinstance(b, SDFCylinder) else 6) for b in bodies], np.int32)
    prm = np.zeros((nb, 3))
    for i, b in enumerate(bodies):
        if isinstance(b, SDFBox):
            prm[i] = b.dims.detach().numpy()
        elif isinstance(b, SDFCylinder):
            prm[i, 0], prm[i, 1] = float(b.rad), float(b.height)
        else:
            prm[i, :2] = latent0
    d["shape_prm"] = prm
    # which bodies carry the reference's analytic mesh (custom_mesh=True); the others its level-set mesh (the default)
    d["custom_mesh"] = np.array([int(bool(getattr(b, "_golden_custom_mesh", False))) for b in bodies])
    d["no_contact"] = np.array([[int(o.geom in b.geom.no_contact) for o in bodies] for b in bodies], np.uint8)
    d["pose0"] = np.stack([b.p.detach().numpy() for b in bodies])
    d["vel0"] = np.stack([b.v.detach().numpy() for b in bodies])
    d["mass"] = np.array([float(b.mass) for b in bodies])
    d["inertia"] = np.stack([b.ang_inertia.detach().numpy() for b in bodies])
    d["restitution"] = np.array([float(b.restitution) for b in bodies])
    d["fric"] = np.array([float(b.fric_coeff) for b in bodies])
    d["fext"] = np.stack([b.apply_forces(0.0).detach().numpy() for b in bodies])
    for i, b in enumerate(bodies):
        d["meshsize_%d" % i] = np.array([len(b.verts), len(b.faces)])
    k = bodies.index(obj)
    d["verts_%d" % k] = obj.verts.detach().numpy()
    d["faces_%d" % k] = obj.faces.numpy().astype(np.int32)
    b0, g0 = contacts_arrays(w.contacts)
    d["init_body"], d["init_geom"] = b0, g0
    stab = []
    while w.t < run_time:
        del _STABLE[:]
        ts = time.time()
        w.step()
        stab.append(np.concatenate(_STABLE) if _STABLE else np.zeros(0, bool))
        print("  t=%.4f nc=%d (%.1f s)" % (w.t, len(w.contacts), time.time() - ts), flush=True)
    T = len(w.trajectory)
    d["traj_t"] = np.array([float(e[0]) for e in w.trajectory])
    d["traj_p"] = np.stack([e[1].detach().numpy().reshape(nb, 7) for e in w.trajectory])
    d["traj_v"] = np.stack([e[2].detach().numpy().reshape(nb, 6) for e in w.trajectory])
    nc = np.array([len(e[3]) for e in w.trajectory], np.int32)
    cb = np.zeros((T, MAXC, 2), np.int32); cg = np.zeros((T, MAXC, 10)); cs = np.zeros((T, MAXC), np.int8)
    for j, e in enumerate(w.trajectory):
        b, g = contacts_arrays(e[3])
        cb[j, :len(b)] = b; cg[j, :len(b)] = g
        if len(stab[j]) == len(b):
            cs[j, :len(b)] = stab[j]
        else:
            cs[j, :len(b)] = -1      # (a step whose last detection was rolled back: not recorded)
    d["traj_nc"], d["traj_body"], d["traj_geom"], d["traj_stable"] = nc, cb, cg, cs
    d["t_final"] = float(w.t)
    loss = (obj.pos - get_tensor(list(target))).norm() ** 2 + 0.05 * latent.norm() ** 2      # demo_meshsdf.py:89
    g, = torch.autograd.grad(loss, [latent])
    d["loss"], d["grad_latent"] = float(loss), g.numpy()
    # the same through a loss on the final position alone (no regulariser), for the stepper's own adjoint
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, "steps", T, "nc", nc.tolist(), "loss", float(loss), "d loss/d latent", g.numpy(), "%.0f s" % (time.time() - t0))


def run_push(name="rollout_igr_push", nsteps=8, seed=0, radius=0.5, latent0=(0.05, -0.03), force0=(3.0, 2.5), mass0=1.0, fric0=0.1):
    """The scene of experiments/system_identification/optim_sysid.py:104-131 (`make_world`: floor, a neural-SDF body of scale 1
    set down on it, gravity, a constant push along x and z, strict_no_penetration=False, fric_dirs=8) with an analytic floor
    mesh, stepped `nsteps` times; loss = sum_t |pos_t - target_t|^2 against a fixed target line (:246-250) and its gradient
    w.r.t. the push, the body's mass and the friction coefficient (shared by both bodies, as there) from torch.autograd."""
    from sdf_physics.physics3d.forces import ExternalForce3D
    from sdf_physics.physics3d.utils import Defaults3D
    t0 = time.time()
    net, _Ws, _bs = fake_igr.seeded_net(seed, radius)
    latent = torch.tensor(latent0, dtype=torch.float64)
    force = torch.tensor(force0, dtype=torch.float64, requires_grad=True)
    mass = torch.tensor([mass0], dtype=torch.float64, requires_grad=True)
    fric = torch.tensor([fric0], dtype=torch.float64, requires_grad=True)

    def force_func(t):
        fv = get_tensor([0, 0, 0, 0, 0, 0])
        fv[[-3, -1]] = force
        return fv
    floor = SDFBox([0, -.5, 0], [20, 1, 20], fric_coeff=fric, restitution=0.0, custom_mesh=True, custom_inertia=True)
    obj = SDF3D([0, 0, 0], scale=1, sdf_func=decode_igr(net), params=[latent], mass=mass, fric_coeff=fric, restitution=0.0)
    obj_pos = get_tensor([0, 0, 0])
    obj_pos[1] = -obj.verts.min(dim=0)[0][1] + 2 * Defaults3D.EPSILON
    obj.set_p(torch.cat([obj_pos.new_ones(1), obj_pos.new_zeros(3), obj_pos]))
    obj.add_force(Gravity3D())
    obj.add_force(ExternalForce3D(force_func))
    w = World3D([floor, obj], [TotalConstraint3D(floor)], time_of_contact_diff=True, strict_no_penetration=False, fric_dirs=8)
    print(name, "world built in %.1f s; body mesh" % (time.time() - t0), len(obj.verts), len(obj.faces), "contacts", len(w.contacts))
    d = dict(igr_seed=seed, igr_radius=radius, latent=np.array(latent0), force=np.array(force0), mass=mass0, fric=fric0, nsteps=nsteps,
             pose0=np.stack([floor.p.detach().numpy(), obj.p.detach().numpy()]), meshsize_1=np.array([len(obj.verts), len(obj.faces)]),
             verts_1=obj.verts.detach().numpy(), faces_1=obj.faces.numpy().astype(np.int32), dt=w.dt)
    target = np.stack([obj.p.detach().numpy()[4:] + np.array([0.02, 0.0, 0.015]) * (k + 1) for k in range(nsteps)])
    loss = 0.0
    for k in range(nsteps):
        w.step(fixed_dt=True)
        loss = loss + ((get_tensor(target[k].tolist()) - obj.pos) ** 2).sum()
        print("  t=%.4f nc=%d" % (w.t, len(w.contacts)), flush=True)
    d["target"] = target
    d["traj_t"] = np.array([float(e[0]) for e in w.trajectory])
    d["traj_p"] = np.stack([e[1].detach().numpy().reshape(2, 7) for e in w.trajectory])
    d["traj_v"] = np.stack([e[2].detach().numpy().reshape(2, 6) for e in w.trajectory])
    d["traj_nc"] = np.array([len(e[3]) for e in w.trajectory], np.int32)
    gf, gm, gc = torch.autograd.grad(loss, [force, mass, fric])
    d["loss"], d["grad_force"], d["grad_mass"], d["grad_fric"] = float(loss), gf.numpy(), gm.numpy(), gc.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, "sub-steps", len(w.trajectory), "nc", d["traj_nc"].tolist(), "loss", float(loss), "grads", gf.numpy(), gm.numpy(), gc.numpy(),
          "%.0f s" % (time.time() - t0))


CASES = {
    "rollout_igr_push": (None, None),
    "rollout_igr_demo": (demo_world, dict(run_time=1.1)),
    "rollout_igr_small": (small_world, dict(run_time=0.4, fixed=(0,), target=(0.0, 1.0, 0.0))),
}


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for name in (sys.argv[1:] or list(CASES)):
        make, kw = CASES[name]
        if make is None:
            run_push(name)
        else:
            run(name, make, **kw)
