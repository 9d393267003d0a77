"""Generate G5/G6 golden rollouts (contacts + trajectories + gradients) from the reference's CPU path.

Run in the build container only:  python -m oracle.gen.gen_rollout_golden
Each tests/golden/rollout_*.npz holds the scene description (incl. the reference's own meshes, so
both sides search identical triangles), the initial contact set, every accepted sub-step of
`World3D.step(fixed_dt=True)` (lcp_physics/physics/world.py:119-139,241-379): t, poses, velocities,
the ordered (body1, body2) list and the contact geometry, and d(sum |pos_T|^2)/d(parameters).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refshim  # noqa: E402

refshim.install()
from oracle.gen import scenes  # noqa: E402
from oracle.gen import contact_record  # noqa: E402

contact_record.install()
from sdf_physics.physics3d.world import World3D  # noqa: E402
from sdf_physics.physics3d.bodies import SDFBowl, SDFBox, SDFBoxRounded, SDFBrick, SDFCylinder, SDFGrid3D, SDFSphere  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
MAXC = 160


def contacts_arrays(contacts):
    n = len(contacts)
    body = np.zeros((n, 2), np.int32); geom = np.zeros((n, 10))
    for k, ((nr, p1, p2, pen), i1, i2) in enumerate(contacts):
        body[k] = (i1, i2)
        geom[k, :3] = nr.detach().numpy(); geom[k, 3:6] = p1.detach().numpy(); geom[k, 6:9] = p2.detach().numpy()
        geom[k, 9] = float(pen)
    return body, geom


def describe(bodies, g=10.0, store_mesh=True):
    d = {}
    nb = len(bodies)
    code = {SDFBox: 0, SDFSphere: 1, SDFCylinder: 2, SDFBoxRounded: 3, SDFBrick: 4, SDFBowl: 5, SDFGrid3D: 7}
    d["shape_type"] = np.array([code[type(b)] for b in bodies], np.int32)
    prm, aux = np.zeros((nb, 3)), np.zeros(nb)
    for i, b in enumerate(bodies):
        if isinstance(b, (SDFBox, SDFBoxRounded, SDFBrick)):
            prm[i] = b.dims.detach().numpy()
            aux[i] = float(getattr(b, "r", 0.0))
        elif isinstance(b, SDFBowl):
            prm[i, 0], prm[i, 1] = float(b.r), float(b.d)
        elif isinstance(b, SDFCylinder):
            prm[i, 0], prm[i, 1] = float(b.rad), float(b.height)
        elif isinstance(b, SDFGrid3D):
            aux[i] = float(b.scale)
            d["grid_%d" % i] = b.sdf.detach().numpy()
        else:
            prm[i, 0] = float(b.rad)
    d["shape_prm"], d["shape_aux"] = prm, aux
    d["no_contact"] = np.array([[int(o.geom in b.geom.no_contact) for o in bodies] for b in bodies], np.uint8)   # add_no_contact (bodies.py:441-445)
    d["pose0"] = np.stack([b.p.detach().numpy() for b in bodies])
    d["vel0"] = np.stack([b.v.detach().numpy() for b in bodies])
    d["mass"] = np.array([float(b.mass) for b in bodies])
    d["inertia"] = np.stack([b.ang_inertia.detach().numpy() for b in bodies])
    d["restitution"] = np.array([float(b.restitution) for b in bodies])
    d["fric"] = np.array([float(b.fric_coeff) for b in bodies])
    d["fext"] = np.stack([b.apply_forces(0.0).detach().numpy() for b in bodies])
    for i, b in enumerate(bodies):
        if not store_mesh and len(b.faces) > 20000:
            # level-set mesh (128^3 marching cubes, ~10^5 faces): the tests rebuild it with the build's own marching cubes
            # -- the stand-in the reference ran with uses the same case tables -- and check these sizes
            d["meshsize_%d" % i] = np.array([len(b.verts), len(b.faces)])
            continue
        d["verts_%d" % i] = b.verts.detach().numpy()
        d["faces_%d" % i] = b.faces.numpy().astype(np.int32)
    return d


def stable_arrays(trajectory):
    st = np.stack([contact_record.lookup(e[3], MAXC)[0] for e in trajectory])
    lap = np.stack([contact_record.lookup(e[3], MAXC)[1] for e in trajectory])
    return st, lap


def branch_b_grads(make, nsteps, toc, jitter=1e-13, **world_kw):
    """The reference's gradient is bimodal on flat-on-flat contacts: `stable_mask = |lap2| < |lap1|`
    (contacts.py:198) compares two rounding-noise Laplacians, so which body's normal carries the gradient
    is decided by the last bit.  A 1e-13 nudge of one initial velocity samples the other branch."""
    bodies, joints, params = make()
    with torch.no_grad():
        bodies[-1].v[3] += jitter
    w = World3D(bodies, joints, time_of_contact_diff=toc, **world_kw)
    init_stable = contact_record.lookup(w.contacts, MAXC)[0]
    for _ in range(nsteps):
        w.step(fixed_dt=True)
    loss = sum((b.pos ** 2).sum() for b in bodies)
    grads = [np.zeros_like(p.detach().numpy()) if g is None else g.numpy()
             for p, g in zip(params, torch.autograd.grad(loss, params, allow_unused=True))]
    return grads, stable_arrays(w.trajectory)[0], init_stable


def run(name, make, nsteps, toc=True, fixed=(0,), store_mesh=True, extra=None, **world_kw):
    bodies, joints, params = make()
    d = describe(bodies, store_mesh=store_mesh)
    d.update(extra(bodies) if extra else {})
    w = World3D(bodies, joints, time_of_contact_diff=toc, **world_kw)
    d["dt"], d["eps"], d["tol"], d["fric_dirs"], d["toc_diff"] = w.dt, w.eps, w.tol, w.fric_dirs, int(toc)
    d["fixed"] = np.array(fixed, np.int32)
    d["strict_no_pen"] = int(w.strict_no_pen)
    d["grad_flags"] = (1 if world_kw.get("stop_contact_grad") else 0) | (2 if world_kw.get("stop_friction_grad") else 0) | \
                      (4 if world_kw.get("detach_contact_b2") else 0)
    b0, g0 = contacts_arrays(w.contacts)
    d["init_body"], d["init_geom"] = b0, g0
    d["init_stable"], d["init_lap"] = contact_record.lookup(w.contacts, MAXC)
    for _ in range(nsteps):
        w.step(fixed_dt=True)
    T = len(w.trajectory)
    nb = len(bodies)
    d["traj_t"] = np.array([float(e[0]) for e in w.trajectory])
    d["traj_p"] = np.stack([e[1].detach().numpy().reshape(nb, 7) for e in w.trajectory])
    d["traj_v"] = np.stack([e[2].detach().numpy().reshape(nb, 6) for e in w.trajectory])
    nc = np.array([len(e[3]) for e in w.trajectory], np.int32)
    cb = np.zeros((T, MAXC, 2), np.int32); cg = np.zeros((T, MAXC, 10))
    for k, e in enumerate(w.trajectory):
        b, g = contacts_arrays(e[3])
        cb[k, :len(b)] = b; cg[k, :len(b)] = g
    d["traj_nc"], d["traj_body"], d["traj_geom"] = nc, cb, cg
    # which body's normal every contact carries (1 = body 2's, 0 = body 1's, -1 = not recorded) and the two Laplacian
    # magnitudes the choice compared (contacts.py:184-198)
    d["traj_stable"], d["traj_lap"] = stable_arrays(w.trajectory)
    d["t_final"] = float(w.t)
    loss = sum((b.pos ** 2).sum() for b in bodies)
    if params:
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        for i, (p, g) in enumerate(zip(params, grads)):
            d["param_%d" % i] = p.detach().numpy()
            d["grad_%d" % i] = np.zeros_like(p.detach().numpy()) if g is None else g.numpy()
        gbs, stB, d["init_stableB"] = branch_b_grads(make, nsteps, toc, **world_kw)
        for i, gb in enumerate(gbs):
            d["gradB_%d" % i] = gb
        d["traj_stableB"] = stB if stB.shape == d["traj_stable"].shape else np.full_like(d["traj_stable"], -1)
    d["loss"] = float(loss)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, "substeps", T, "for", nsteps, "steps; nc range", nc.min(), nc.max(), "loss", float(loss),
          "grads", [np.abs(d["grad_%d" % i]).max() for i in range(len(params))])


CASES = {
    "rollout_sphere": (lambda: scenes.sphere_drop(seed=1, floor_dims=(4.0, 1.0, 4.0)), dict(nsteps=24)),
    "rollout_sphere_notoc": (lambda: scenes.sphere_drop(seed=1, floor_dims=(4.0, 1.0, 4.0)), dict(nsteps=24, toc=False)),
    "rollout_stack1": (lambda: scenes.box_stack(nbox=1, seed=3, vel_scale=1.0, push=2.0), dict(nsteps=4)),
    "rollout_stack2": (lambda: scenes.box_stack(nbox=2, seed=4, vel_scale=0.5, push=1.5), dict(nsteps=3)),
    # BASELINE configs[2] itself at batch 1: floor + 7 stacked boxes (48 velocities, ~560 inequality rows), shoved so that
    # friction saturates and the gradients are non-zero
    "rollout_stack7": (lambda: scenes.box_stack(nbox=7, seed=11, vel_scale=0.3, push=1.0), dict(nsteps=3)),
    "rollout_boxdrop": (lambda: scenes.box_drop(seed=7), dict(nsteps=12)),
    "rollout_cylinder": (lambda: scenes.cylinder_drop(seed=9), dict(nsteps=10)),
    # long horizon: 100 outer steps, 245 sub-steps, several bounces with time-of-contact events, coming to rest
    "rollout_sphere_long": (lambda: scenes.sphere_drop(seed=1, floor_dims=(4.0, 1.0, 4.0)), dict(nsteps=100)),
    "rollout_bigbox": (lambda: scenes.big_box(), dict(nsteps=3)),
    # cases that came out of random comparisons against the reference (tools/dbg_fuzz.py): each one exposed a difference
    # in the thinning stage's hull (coincident points, a mid-edge start vertex, Qhull's merge tolerance) or exercises
    # bookkeeping no other golden has (pinned body not first, no_contact pairs, sphere against sphere)
    "rollout_boxdrop_fd4": (lambda: scenes.box_drop(seed=7), dict(nsteps=12, fric_dirs=4)),      # World3D(fric_dirs=4)
    "rollout_two_spheres": (lambda: scenes.two_spheres(), dict(nsteps=12)),
    "rollout_sphere_on_box": (lambda: scenes.sphere_on_box(), dict(nsteps=12)),
    "rollout_floor_last": (lambda: scenes.floor_last(), dict(nsteps=20, fixed=(1,))),
    "rollout_no_contact": (lambda: scenes.no_contact_pair(), dict(nsteps=6)),
    "rollout_sphere_roll": (lambda: scenes.sphere_drop(seed=60), dict(nsteps=40)),
    # the remaining primitives (bodies.py:857-885).  SDFBowl has no rollout golden: the reference's bowl evaluates its
    # normal at a point shifted twice (in-place shift in both bowl_sdf and bowl_sdf_grad, bodies.py:99,119), and its own
    # World3D.step does not get past the first sphere-in-bowl contact (dt halving without end); the bowl is pinned at the
    # query level (sdf_query.npz).
    "rollout_rounded": (lambda: scenes.rounded_drop("rounded"), dict(nsteps=10, store_mesh=False)),
    # the same scene with d sum|pos_T|^2 / d dims: through the SDF, through the level-set mesh (MeshSDF backward) and
    # through the inertia integrated over that mesh
    "rollout_rounded_grad": (lambda: scenes.rounded_drop("rounded", requires_grad=True), dict(nsteps=10, store_mesh=False)),
    "rollout_levelset_sphere": (lambda: scenes.levelset_sphere(), dict(nsteps=12, store_mesh=False)),
    "rollout_levelset_cylinder": (lambda: scenes.levelset_cylinder(), dict(nsteps=8, store_mesh=False)),
    "rollout_levelset_box": (lambda: scenes.levelset_box(), dict(nsteps=3, store_mesh=False)),
    # World3D's gradient switches (physics3d/world.py:33-37) on the tilted box drop: same trajectory, other gradients
    "rollout_boxdrop_stop_contact": (lambda: scenes.box_drop(seed=7), dict(nsteps=12, stop_contact_grad=True)),
    "rollout_boxdrop_stop_friction": (lambda: scenes.box_drop(seed=7), dict(nsteps=12, stop_friction_grad=True)),
    "rollout_boxdrop_detach_b2": (lambda: scenes.box_drop(seed=7), dict(nsteps=12, detach_contact_b2=True)),
    # level-set rounded box at rest on a rounded-rimmed side: a true 3-D hull of thousands of nearly coplanar points
    "rollout_rounded_rest": (lambda: scenes.rounded_rest(), dict(nsteps=4, store_mesh=False)),
    "rollout_brick": (lambda: scenes.rounded_drop("brick"), dict(nsteps=10, store_mesh=False)),
    # strict_no_penetration=False, and a sphere too fast for any halving of dt to catch in the contact band: the escape of
    # world.py:345-347 (dt < dt/2^10: go on with the penetrating contacts, unthinned, no time-of-contact bookkeeping)
    "rollout_fast_sphere": (lambda: scenes.fast_sphere(), dict(nsteps=3, strict_no_penetration=False)),
    # a voxel-grid SDF body (SDFGrid3D): trilinear samples, central-difference normals; gradient w.r.t. its start velocity
    "rollout_grid_body": (lambda: scenes.grid_body_drop(), dict(nsteps=12)),
}


def main():
    """python -m oracle.gen.gen_rollout_golden [case ...]   (default: all)"""
    os.makedirs(OUT, exist_ok=True)
    for name in (sys.argv[1:] or list(CASES)):
        make, kw = CASES[name]
        run(name, make, **kw)


if __name__ == "__main__":
    main()
