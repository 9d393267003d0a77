"""Batched experiment drivers: the optimisation loops of the reference's `experiments/`, for a whole
batch of independent scenes at once, without sacred / tensorboard / pyrender.

trajectory fitting  (experiments/trajectory_fitting/optim_sphere.py)
    bounce_world(radii, ...)                      floor + wall + sphere thrown at the wall (:77-111), one scene per radius
    run_world_fixed_dt(world, run_time, detach_2nd_bounce)   fixed outer steps until every scene has reached run_time; with
                                                  detach_2nd_bounce the second step in a row that ends in contact is undone
                                                  and redone with poses and velocities cut from the graph (:163-177), per scene
    trajectory_loss(traj, traj_target)            mean over a scene's entries of |pos - pos_target|^2 at the nearest target time (:114-160)
    chamfer(a, b)                                 symmetric mean squared nearest-neighbour distance (pytorch3d.loss.chamfer_distance, :244)
    fit_sphere_radius(...)                        the gradient-descent loop (:210-270) for B (target, start) pairs; radius_error_table()
                                                  prints min / mean / max |r - r*| like RESULTS.md:14-47
trajectory fitting in shape space  (experiments/trajectory_fitting/optim_shapespace.py)
    bounce_world_latent(latents, packed, ...)     the same scene with a neural-SDF body in place of the sphere (:90-124)
    fit_trajectory_latent(...)                    the loop over the latent code
inertia fitting  (experiments/inertia_fitting/optim_shapespace.py)
    spin_world(latents, torque_dirs, net)         one neural-SDF body per scene, translation locked (X/Y/Z constraints), torque for
                                                  t < 0.3 (:71-92); inertia from the body's level-set mesh, differentiable w.r.t. the latent
    fit_inertia_latent(...)                       the loop of :136-250: loss = |v_T - v_T*|^2 + reg |latent|^2
inertia fitting of primitives  (experiments/inertia_fitting/optim_primitives.py)
    primitive_spin_world(kind, dims)              the same scene for an SDFBox / SDFSphere / SDFCylinder per row of dimensions (:95-115)
    fit_inertia_primitive(kind, ...)              the loop of :160-240 (Adam, dimensions clamped to [0.5, 2])

system identification  (experiments/system_identification/optim_sysid.py)
    push_world(latents, packed, force, mass, fric, ...)   the floor and a neural-SDF body pushed along it (:104-131), per scene its own
                                                  push, mass and friction coefficient (torch tensors that may require grad)
    fit_sysid(goal, ...)                          the loop of :184-300 for goal in ('mass', 'force', 'friction')

    python -m diffsdfsim_amd.experiments sphere --scenes 64 --iters 100
    python -m diffsdfsim_amd.experiments shapespace --scenes 4 --iters 10
    python -m diffsdfsim_amd.experiments inertia --scenes 8 --iters 10
    python -m diffsdfsim_amd.experiments primitives --kind box --scenes 8 --iters 50
    python -m diffsdfsim_amd.experiments sysid --goal mass --scenes 8 --iters 20
"""
import argparse
import math

import numpy as np
import torch

from . import mass_properties, meshes, meshsdf, scenes
from . import world_abi as abi
from .physics3d import BatchWorld3D
from .physics3d.utils import Defaults3D


# ---- rollouts ------------------------------------------------------------------------------------------------------------
def rollout(world, steps):
    """`steps` fixed outer steps of every scene -> poses [T,B,nb,7], velocities [T,B,nb,6] (autograd-connected)."""
    P, V = [], []
    for _ in range(steps):
        world.step()
        P.append(world.pose)
        V.append(world.vel)
    return torch.stack(P), torch.stack(V)


def run_world_fixed_dt(world, run_time, detach_2nd_bounce=False, max_steps=10000):
    """`run_world_fixed_dt` (optim_sphere.py:163-177) per scene of a batch.  Returns the trajectory as the reference's
    `world.trajectory` holds it, as a dict of t [K,B], pose [K,B,nb,7], vel [K,B,nb,6], valid [K,B]: ONE ENTRY PER ACCEPTED
    SUB-STEP (a step that halves dt or goes through a time-of-contact event adds several), stamped with the time at the START
    of the sub-step and holding the state after it (world.py:373-379: the entry is appended before `self.t += dt`); every entry
    keeps its graph.  An undone step (detach_2nd_bounce) leaves the list -- except its first entry: `undo_step` pops
    `while self.trajectory[-1][0] > self.t` (world.py:114-116) and that entry's stamp EQUALS self.t, so it stays, un-detached,
    as it does in the reference."""
    B, dev = world.B, world.device
    world.record_substeps = True
    num_contact = np.zeros(B, np.int64)
    T, P, V, OK = [], [], [], []
    for _ in range(max_steps):
        t = world.t
        mask = t < run_time
        if not mask.any():
            break
        had = np.asarray(world.step(mask=mask)) & mask
        sub, pose_end, vel_end = world.substeps, world.pose, world.vel
        redo = np.zeros(B, bool)
        if detach_2nd_bounce:
            num_contact += had
            redo = had & (num_contact > 1)
            if redo.any():
                world.undo_step(redo)
                world.detach_state(redo)
                num_contact[redo] = 0
        m_t = torch.as_tensor(mask, device=dev); r_t = torch.as_tensor(redo, device=dev)
        K = int(sub["pose"].shape[0])
        for k in range(K):           # the sub-steps inside this outer step
            T.append(sub["t"][k]); P.append(sub["pose"][k]); V.append(sub["vel"][k])
            OK.append(sub["valid"][k] & m_t & (~r_t if k > 0 else torch.ones_like(r_t)))
        # the last sub-step's entry = the state the step ended in; of an undone step it survives only if it is the step's FIRST entry
        first_is_last = ~sub["valid"][0] if K > 0 else torch.ones(B, dtype=torch.bool, device=dev)
        T.append(sub["last_t"]); P.append(pose_end); V.append(vel_end)
        OK.append(m_t & (~r_t | first_is_last))
    return dict(t=torch.stack(T), pose=torch.stack(P), vel=torch.stack(V), valid=torch.stack(OK))


def trajectory_loss(traj, target, body=-1):
    """optim_sphere.py:114-160: every entry of a scene's trajectory is compared with the target entry nearest in time (position
    of the last body); the sum over the scene's entries divided by their number.  Of two target entries equally near the LATER
    one is taken (`if diff <= min_diff`, :131).  -> [B]"""
    tw, tt = traj["t"], target["t"].to(traj["t"])
    diff = (tw[:, None, :] - tt[None, :, :]).abs()                                                             # [Kw, Kt, B]
    diff = torch.where(target["valid"].to(tw.device)[None, :, :], diff, torch.full_like(diff, float("inf")))
    Kt = diff.shape[1]
    j = Kt - 1 - diff.flip(1).argmin(dim=1)                                                                     # [Kw, B], last of the minima
    pos_t = target["pose"][:, :, body, 4:].to(traj["pose"])
    near = torch.gather(pos_t, 0, j[:, :, None].expand(-1, -1, 3))
    d = traj["pose"][:, :, body, 4:] - near
    w = traj["valid"].to(d.dtype)
    return ((d * d).sum(dim=2) * w).sum(dim=0) / w.sum(dim=0).clamp_min(1.0)


def chamfer(a, b):
    """pytorch3d.loss.chamfer_distance of two point sets a [Na,3], b [Nb,3] (defaults: squared distances, mean over the points of
    each set, the two directions summed)."""
    # on the device, a block of rows at a time: level-set meshes have 10^4 - 10^5 vertices
    dev = torch.device("cuda") if torch.cuda.is_available() and not a.is_cuda else a.device
    a, b = a.to(dev), b.to(dev)
    to_b = torch.full((b.shape[0],), float("inf"), dtype=a.dtype, device=dev)
    to_a = a.new_zeros(())
    for i in range(0, a.shape[0], 4096):
        d = torch.cdist(a[i:i + 4096], b) ** 2
        to_a = to_a + d.min(dim=1).values.sum()
        to_b = torch.minimum(to_b, d.min(dim=0).values)
    return to_a / a.shape[0] + to_b.mean()


# ---- trajectory fitting: a sphere thrown at a wall (optim_sphere.py) --------------------------------------------------------
def bounce_world(radii, use_toc_diff=True, use_friction=True, use_wall=True, use_floor=True, use_gravity=True,
                 sphere_pos=(0.0, 5.0, 0.0), sphere_vel=(5.0, 0.0, 0.0), run_time=1.5, dt=Defaults3D.DT, device=None):
    """optim_sphere.py:77-111 for one scene per radius: floor 20 x 1 x 20, wall [5,5,0] 1 x 10 x 10 (both pinned, no contact
    with each other), a sphere of the scene's radius at (0,5,0) thrown with (5,0,0); restitution 0.5, friction 0.25 (0 without).
    Analytic meshes and inertias (the reference's custom_mesh / custom_inertia); shape and inertia are functions of `radii`."""
    rad = np.asarray(radii.detach().cpu() if torch.is_tensor(radii) else radii, np.float64).reshape(-1)
    B = len(rad)
    fixed = [n for n, on in (("floor", use_floor), ("wall", use_wall)) if on]
    nb = len(fixed) + 1
    mu = 0.25 if use_friction else 0.0
    spec, cache = scenes._base(B, nb), {}
    spec["Je"] = np.zeros((B, 6 * len(fixed), 6 * nb))
    nocon = np.zeros((nb, nb), np.uint8)
    for k, name in enumerate(fixed):
        dims, pos = ((20.0, 1.0, 20.0), (0.0, -0.5, 0.0)) if name == "floor" else ((1.0, 10.0, 10.0), (5.0, 5.0, 0.0))

        def make(dims=dims):
            v, f, tie = meshes.box_mesh(np.asarray(dims))
            return v, f, 0.5 * tie
        spec["mesh_id"][:, k] = scenes._add_mesh(spec, cache, ("box",) + tuple(dims), make)
        spec["pose"][:, k, 4:] = pos
        spec["shape_prm"][:, k] = dims
        spec["inertia"][:, k] = scenes.box_inertia(1.0, np.asarray(dims))
        spec["fric"][:, k] = mu
        spec["restitution"][:, k] = Defaults3D.RESTITUTION
        spec["Je"][:, 6 * k:6 * k + 6, 6 * k:6 * k + 6] = np.eye(6)
    if len(fixed) == 2:
        nocon[0, 1] = nocon[1, 0] = 1
    spec["no_contact"] = nocon
    uv, uf = meshes.icosphere(4)
    s_ = nb - 1
    for s in range(B):
        spec["mesh_id"][s, s_] = scenes._add_mesh(spec, cache, ("sphere", float(rad[s])), lambda r=float(rad[s]): (uv * r, uf, uv.copy()))
    spec["shape_type"][:, s_] = abi.SHAPE_SPHERE
    spec["shape_prm"][:, s_, 0] = rad
    spec["pose"][:, s_, 4:] = sphere_pos
    spec["vel"][:, s_, 3:] = sphere_vel
    spec["inertia"][:, s_] = 0.4 * rad[:, None, None] ** 2 * np.eye(3)
    spec["fric"][:, s_] = mu
    spec["restitution"][:, s_] = Defaults3D.RESTITUTION
    if use_gravity:
        spec["fext"][:, s_, 4] = -10.0
    params = {}
    if torch.is_tensor(radii) and radii.requires_grad:
        r = radii.to(torch.float64).reshape(-1)
        prm = torch.tensor(spec["shape_prm"], dtype=torch.float64)
        prm = torch.cat([prm[:, :s_], torch.stack([r, torch.zeros_like(r), torch.zeros_like(r)], 1)[:, None].to(prm)], 1)
        inertia = torch.tensor(spec["inertia"], dtype=torch.float64)
        ball = (0.4 * r * r)[:, None, None].to(inertia) * torch.eye(3, dtype=torch.float64)      # 2/5 m r^2 (bodies.py:993-994)
        params = dict(shape_prm=prm, inertia=torch.cat([inertia[:, :s_], ball[:, None]], 1))
    steps = int(math.ceil(run_time / dt)) + 2
    return BatchWorld3D(spec, params=params, dt=dt, time_of_contact_diff=use_toc_diff, max_substeps=8 * steps + 64, device=device)


def fit_sphere_radius(target_radii, start_radii, run_time=1.5, max_iter=100, lr=0.1, conv_thresh=1e-5, min_dim=0.4, max_dim=2.0,
                      detach_2nd_bounce=True, log=None, **scene):
    """The loop of optim_sphere.py:210-270 for B scenes at once: plain gradient descent on the radius, a new world from the
    current estimate in every iteration, a scene stops when its loss changes by less than conv_thresh (its radius is frozen:
    the scenes of the batch are independent optimisations).  Returns dict(radius [B], history)."""
    target_radii = torch.as_tensor(np.asarray(target_radii, np.float64))
    with torch.no_grad():
        target = run_world_fixed_dt(bounce_world(target_radii, run_time=run_time, **scene), run_time)
    rad = torch.tensor(np.asarray(start_radii, np.float64), requires_grad=True)
    B = rad.numel()
    done = np.zeros(B, bool)
    last = np.full(B, 1e10)
    hist = []
    for e in range(max_iter):
        if rad.grad is not None:
            rad.grad = None
        world = bounce_world(rad, run_time=run_time, **scene)
        traj = run_world_fixed_dt(world, run_time, detach_2nd_bounce=detach_2nd_bounce)
        loss = trajectory_loss(traj, target)
        loss.sum().backward()
        l = loss.detach().cpu().numpy()
        done |= np.abs(last - l) < conv_thresh
        hist.append(dict(iter=e, loss=l.copy(), radius=rad.detach().numpy().copy(), grad=rad.grad.numpy().copy(), done=done.copy()))
        if log:
            log("iter %3d  mean loss %.3e  mean |r - r*| %.4f  converged %d / %d" %
                (e, float(l.mean()), float((rad.detach() - target_radii).abs().mean()), int(done.sum()), B))
        if done.all():
            break
        with torch.no_grad():
            step = lr * rad.grad
            step[torch.as_tensor(done)] = 0.0
            rad -= step
            rad.clamp_(min_dim, max_dim)
        last = l
    return dict(radius=rad.detach().numpy().copy(), target=target_radii.numpy(), history=hist)


def radius_error_table(results):
    """min / mean / max of |r - r*| per variant, the numbers of RESULTS.md:14-47."""
    rows = []
    for name, r in results.items():
        err = np.abs(r["radius"] - r["target"])
        rows.append("%-22s min %.1e  mean %.4f  max %.4f   (%d scenes)" % (name, err.min(), err.mean(), err.max(), len(err)))
    return "\n".join(rows)


# ---- trajectory fitting in shape space: a neural-SDF body thrown at a wall (trajectory_fitting/optim_shapespace.py) ----------
def bounce_world_latent(latents, packed, use_toc_diff=True, use_friction=True, use_wall=True, use_floor=True, use_gravity=False,
                        obj_pos=(0.0, 5.0, 0.0), obj_vel=(5.0, 0.0, 0.0), run_time=1.5, dt=Defaults3D.DT, res=128, device=None):
    """trajectory_fitting/optim_shapespace.py:90-124 for one scene per latent code: floor 50 x 1 x 50, wall [5,5,0] 1 x 10 x 10
    (pinned, no contact with each other), the neural-SDF body (scale 1, mass 1) at (0,5,0) thrown with (5,0,0); restitution
    0.5, friction 0.25; no gravity by default, strict_no_penetration=False.  With `latents` a tensor that requires grad the
    scene is differentiable w.r.t. it three ways, as the reference's is: the SDF the contacts query, the level-set mesh
    (MeshSDF) the contact candidates come from, and the inertia integrated over that mesh."""
    lat_t = torch.as_tensor(latents, dtype=torch.float64)
    lat = lat_t.detach().cpu().numpy()
    B = lat.shape[0]
    fixed = [n for n, on in (("floor", use_floor), ("wall", use_wall)) if on]
    nb = len(fixed) + 1
    mu = 0.25 if use_friction else 0.0
    spec, cache = scenes._base(B, nb), {}
    spec["Je"] = np.zeros((B, 6 * len(fixed), 6 * nb))
    nocon = np.zeros((nb, nb), np.uint8)
    for k, name in enumerate(fixed):
        dims, pos = ((50.0, 1.0, 50.0), (0.0, -0.5, 0.0)) if name == "floor" else ((1.0, 10.0, 10.0), (5.0, 5.0, 0.0))

        def make(dims=dims):
            v, f, tie = meshes.box_mesh(np.asarray(dims))
            return v, f, 0.5 * tie
        spec["mesh_id"][:, k] = scenes._add_mesh(spec, cache, ("box",) + tuple(dims), make)
        spec["pose"][:, k, 4:] = pos
        spec["shape_prm"][:, k] = dims
        spec["inertia"][:, k] = scenes.box_inertia(1.0, np.asarray(dims))
        spec["fric"][:, k] = mu
        spec["restitution"][:, k] = Defaults3D.RESTITUTION
        spec["Je"][:, 6 * k:6 * k + 6, 6 * k:6 * k + 6] = np.eye(6)
    if len(fixed) == 2:
        nocon[0, 1] = nocon[1, 0] = 1
    spec["no_contact"] = nocon
    s_ = nb - 1
    spec["shape_aux"] = np.zeros((B, nb))
    spec["igr_net"] = packed
    spec["shape_type"][:, s_] = abi.SHAPE_IGR
    spec["shape_aux"][:, s_] = 1.0
    one = torch.tensor(1.0, dtype=torch.float64)
    vts, Is = [torch.as_tensor(m[0], dtype=torch.float64) for m in spec["meshes"]], []
    for s in range(B):
        v, f = meshsdf.igr_mesh(lat_t[s], packed, res=res)
        Is.append(mass_properties.mesh_inertia_diff(v, f, one).cpu())
        spec["meshes"].append((v.detach().cpu().numpy(), f.cpu().numpy())); spec["mesh_vgrad"].append(np.zeros((v.shape[0], 3)))
        spec["mesh_id"][s, s_] = len(spec["meshes"]) - 1
        vts.append(v)
    spec["shape_prm"][:, s_, :2] = lat
    spec["pose"][:, s_, 4:] = obj_pos
    spec["vel"][:, s_, 3:] = obj_vel
    spec["inertia"][:, s_] = torch.stack(Is).detach().numpy()
    spec["fric"][:, s_] = mu
    spec["restitution"][:, s_] = Defaults3D.RESTITUTION
    if use_gravity:
        spec["fext"][:, s_, 4] = -10.0
    params = {}
    if lat_t.requires_grad:
        prm = torch.tensor(spec["shape_prm"], dtype=torch.float64)
        mine = torch.cat([lat_t.cpu(), torch.zeros(B, 1, dtype=torch.float64)], 1)[:, None]
        inertia = torch.tensor(spec["inertia"], dtype=torch.float64)
        dev = vts[-1].device
        params = dict(shape_prm=torch.cat([prm[:, :s_], mine], 1), inertia=torch.cat([inertia[:, :s_], torch.stack(Is)[:, None]], 1),
                      verts=torch.cat([v.to(dev) for v in vts]))
    steps = int(math.ceil(run_time / dt)) + 2
    return BatchWorld3D(spec, params=params, dt=dt, time_of_contact_diff=use_toc_diff, strict_no_penetration=False,
                        max_substeps=8 * steps + 64, device=device, maxc=256, max_cand=8192, max_pc=128)


def fit_trajectory_latent(target_latents, start_latents, packed, run_time=1.0, max_iter=100, lr=1e-3, conv_thresh=1e-5,
                          latent_reg=1e-4, optimizer="Adam", detach_2nd_bounce=True, log=None, **scene):
    """trajectory_fitting/optim_shapespace.py:232-320 for B scenes at once (its settings: Adam, lr 1e-3, run_time 1,
    latent_reg 1e-4): the latent code is fitted so that the thrown body's trajectory (its positions at the nearest target
    times, trajectory_loss) matches the target's; loss = trajectory_loss + latent_reg |latent|^2."""
    with torch.no_grad():
        target = run_world_fixed_dt(bounce_world_latent(torch.as_tensor(np.asarray(target_latents, np.float64)), packed,
                                                        run_time=run_time, **scene), run_time)
    lat = torch.tensor(np.asarray(start_latents, np.float64), requires_grad=True)
    opt = torch.optim.Adam([lat], lr=lr) if optimizer == "Adam" else torch.optim.SGD([lat], lr=lr)
    hist, last = [], None
    for e in range(max_iter):
        opt.zero_grad()
        world = bounce_world_latent(lat, packed, run_time=run_time, **scene)
        traj = run_world_fixed_dt(world, run_time, detach_2nd_bounce=detach_2nd_bounce)
        loss = trajectory_loss(traj, target)
        loss = loss + latent_reg * (lat.to(loss.device) ** 2).sum(dim=1)
        loss.sum().backward()
        l = loss.detach().cpu().numpy()
        hist.append(dict(iter=e, loss=l.copy(), latent=lat.detach().numpy().copy(), grad=lat.grad.numpy().copy()))
        if log:
            log("iter %3d  mean loss %.3e  mean |latent - target| %.4f  |grad| %.3e" %
                (e, float(l.mean()), float(np.abs(lat.detach().numpy() - np.asarray(target_latents)).mean()), float(lat.grad.abs().mean())))
        if last is not None and np.all(np.abs(last - l) < conv_thresh):
            break
        opt.step()
        last = l
    return dict(latent=lat.detach().numpy().copy(), target=np.asarray(target_latents), history=hist)


# ---- inertia fitting: a neural-SDF body spun by a torque (optim_shapespace.py) ------------------------------------------------
def spin_world(latents, torque_dirs, packed, scale=1.0, mass=1.0, res=128, steps=64, device=None, keep_meshes=True):
    """optim_shapespace.py:71-92 for one scene per latent code: a single neural-SDF body (scale 1) whose translation is locked
    by X/Y/Z constraints; no contacts, so only its inertia matters -- integrated over its level-set mesh (MeshSDF:
    differentiable w.r.t. the latent).  The torque (t < 0.3) is applied by the caller through world.params['fext']."""
    B = latents.shape[0]
    Is, ms = [], []
    for s in range(B):
        v, f = meshsdf.igr_mesh(latents[s], packed, res=res)
        vt = v * scale
        Is.append(mass_properties.mesh_inertia_diff(vt, f, torch.tensor(float(mass), dtype=torch.float64)))
        # (the meshes are only kept for the chamfer distance of the experiment's report; a caller that does not need them saves
        # a device -> host copy of ~1 MB and a synchronisation per scene)
        ms.append((vt.detach().cpu().numpy(), f.cpu().numpy()) if keep_meshes else (None, int(f.shape[0])))
    return _spin_world(torch.stack(Is), ms, scale, mass, steps, device)


def _spin_world(inertias, ms, scale, mass, steps, device):
    """One free-spinning body per scene with the given body-frame inertia tensors [B, 3, 3] (graph kept) and meshes."""
    B = len(ms)
    inertia = inertias[:, None]                                               # [B,1,3,3], graph to the shape parameters
    one = lambda a: np.tile(np.asarray(a, np.float64), (B, 1, 1))
    spec = dict(pose=one([1.0, 0, 0, 0, 0, 0, 0]), vel=one(np.zeros(6)), mass=np.full((B, 1), float(mass)),
                inertia=inertia.detach().cpu().numpy(), restitution=np.zeros((B, 1)), fric=np.zeros((B, 1)), fext=np.zeros((B, 1, 6)),
                shape_type=np.full((B, 1), abi.SHAPE_SPHERE, np.int32), shape_prm=one([scale, 0, 0]),      # (nothing collides: the shape is never queried)
                # (a lone body per scene: the stepper never searches its mesh, so it gets one shared triangle instead of B
                # level-set meshes of 10^5 faces whose search structures would take seconds to build; the real meshes stay on
                # the world object for the chamfer distance)
                mesh_id=np.zeros((B, 1), np.int32), meshes=[(0.1 * np.eye(3), np.array([[0, 1, 2]]))], mesh_vgrad=[np.zeros((3, 3))],
                Je=np.tile(np.concatenate([np.zeros((3, 3)), np.eye(3)], 1), (B, 1, 1)), no_contact=np.zeros((1, 1), np.uint8))
    w = BatchWorld3D(spec, params=dict(inertia=inertia), max_substeps=steps + 16, device=device)
    w.meshes = ms
    return w


PRIMITIVES = {"box": 3, "sphere": 1, "cylinder": 2}      # parameters per kind (optim_primitives.py:77-92)


def primitive_spin_world(kind, dims, mass=1.0, steps=64, device=None):
    """optim_primitives.py:95-115 for one scene per parameter row: an SDFBox / SDFSphere / SDFCylinder at the origin with the
    reference's defaults there (custom_mesh = custom_inertia = False: marching-cubes mesh of the analytic SDF, inertia from that
    mesh, differentiable w.r.t. the dimensions through the mesher), translation locked."""
    from .physics3d import SDFBox, SDFCylinder, SDFSphere
    make = {"box": lambda d: SDFBox([0, 0, 0], d, mass=mass, custom_mesh=False, custom_inertia=False),
            "sphere": lambda d: SDFSphere([0, 0, 0], d[0], mass=mass, custom_mesh=False, custom_inertia=False),
            "cylinder": lambda d: SDFCylinder([0, 0, 0], rad=d[0], height=d[1], mass=mass, custom_mesh=False, custom_inertia=False)}[kind]
    bodies = [make(dims[s]) for s in range(dims.shape[0])]
    ms = [(b.verts_np, np.asarray(b.faces_np)) for b in bodies]
    return _spin_world(torch.stack([b.ang_inertia.to(torch.float64).cpu() for b in bodies]), ms, 1.0, mass, steps, device)


def run_spin(world, torque_dirs, run_time=2.0, torque_time=0.3):
    """`run_world(world, run_time=...)` with the experiment's force function: torque_dir for t < 0.3, then nothing."""
    B = world.B
    tq = torch.cat([torch.as_tensor(torque_dirs, dtype=torch.float64), torch.zeros(B, 3, dtype=torch.float64)], 1)[:, None]
    while float(world.t.min()) < run_time:
        world.params["fext"] = tq if float(world.t.min()) < torque_time else torch.zeros_like(tq)
        world.step()
    return world.vel[:, 0]


def fit_inertia_latent(target_latents, start_latents, torque_dirs, packed, run_time=2.0, max_iter=20, lr=1e-2, latent_reg=0.0,
                       conv_thresh=1e-7, res=128, log=None):
    """optim_shapespace.py:136-250: gradient descent on the latent code so that the body's final velocity under the torque
    matches the target's.  The gradient runs latent -> level-set mesh (MeshSDF) -> volume integrals -> world-frame inertia ->
    linear solve of every step."""
    steps = int(math.ceil(run_time / Defaults3D.DT)) + 2
    with torch.no_grad():
        wt = spin_world(torch.as_tensor(target_latents, dtype=torch.float64), torque_dirs, packed, res=res, steps=steps)
        v_target = run_spin(wt, torque_dirs, run_time).clone()
    lat = torch.tensor(np.asarray(start_latents, np.float64), requires_grad=True)
    hist, last = [], None
    for e in range(max_iter):
        if lat.grad is not None:
            lat.grad = None
        w = spin_world(lat, torque_dirs, packed, res=res, steps=steps)
        v = run_spin(w, torque_dirs, run_time)
        loss = ((v - v_target.to(v)) ** 2).sum(dim=1) + latent_reg * (lat.to(v.device) ** 2).sum(dim=1)
        loss.sum().backward()
        l = loss.detach().cpu().numpy()
        d = np.array([float(chamfer(torch.as_tensor(a[0]), torch.as_tensor(b[0]))) for a, b in zip(w.meshes, wt.meshes)])
        hist.append(dict(iter=e, loss=l.copy(), latent=lat.detach().numpy().copy(), grad=lat.grad.numpy().copy(), chamfer=d))
        if log:
            log("iter %3d  mean loss %.3e  mean chamfer %.3e  |grad| %.3e" % (e, float(l.mean()), float(d.mean()), float(lat.grad.abs().mean())))
        if last is not None and np.all(np.abs(last - l) < conv_thresh):
            break
        with torch.no_grad():
            lat -= lr * lat.grad
        last = l
    return dict(latent=lat.detach().numpy().copy(), target=np.asarray(target_latents), history=hist)


def fit_inertia_primitive(kind, target_dims, start_dims, torque_dirs, run_time=2.0, max_iter=200, lr=1e-2, conv_thresh=1e-5,
                          min_dim=0.5, max_dim=2.0, optimizer="Adam", log=None):
    """optim_primitives.py:160-240 for B (target, start) rows at once: Adam (or plain gradient descent) on the dimensions so
    that the final velocity under the torque matches the target's; dimensions clamped to [min_dim, max_dim] after each update;
    stops when no scene's loss moved by more than conv_thresh."""
    steps = int(math.ceil(run_time / Defaults3D.DT)) + 2
    tgt = torch.as_tensor(np.asarray(target_dims, np.float64)).reshape(-1, PRIMITIVES[kind])
    with torch.no_grad():
        wt = primitive_spin_world(kind, tgt, steps=steps)
        v_target = run_spin(wt, torque_dirs, run_time).clone()
    dims = torch.tensor(np.asarray(start_dims, np.float64).reshape(-1, PRIMITIVES[kind]), requires_grad=True)
    opt = torch.optim.Adam([dims], lr=lr) if optimizer == "Adam" else torch.optim.SGD([dims], lr=lr)
    hist, last = [], None
    for e in range(max_iter):
        opt.zero_grad()
        w = primitive_spin_world(kind, dims, steps=steps)
        v = run_spin(w, torque_dirs, run_time)
        loss = ((v - v_target.to(v)) ** 2).sum(dim=1)
        loss.sum().backward()
        l = loss.detach().cpu().numpy()
        d = np.array([float(chamfer(torch.as_tensor(a[0]), torch.as_tensor(b[0]))) for a, b in zip(w.meshes, wt.meshes)])
        hist.append(dict(iter=e, loss=l.copy(), dims=dims.detach().numpy().copy(), grad=dims.grad.numpy().copy(), chamfer=d))
        if log:
            log("iter %3d  mean loss %.3e  mean chamfer %.3e  mean |dims - target| %.4f" %
                (e, float(l.mean()), float(d.mean()), float((dims.detach() - tgt).abs().mean())))
        if last is not None and np.all(np.abs(last - l) < conv_thresh):
            break
        opt.step()
        with torch.no_grad():
            dims.clamp_(min_dim, max_dim)
        last = l
    return dict(dims=dims.detach().numpy().copy(), target=tgt.numpy(), history=hist)


# ---- system identification (experiments/system_identification/optim_sysid.py) --------------------------------------------
def push_world(latents, packed, force, mass, fric, run_steps, floor_dims=(20.0, 1.0, 20.0), restitution=0.0, g=10.0, res=128, device=None,
               mesh_cache=None):
    """optim_sysid.py:104-131 (`make_world`) for one scene per latent code: the floor and a neural-SDF body (scale 1) set down
    on it (2 eps above, by its mesh's lowest vertex), gravity, a constant push (force[:, 0] along x, force[:, 1] along z),
    strict_no_penetration=False, fric_dirs=8.  `force` [B,2], `mass` [B], `fric` [B] (both bodies, as in the experiment) are
    torch tensors and may require grad: the batch's parameters are built from them (inertia = mass x the mesh's unit inertia)."""
    latents = np.asarray(latents, np.float64)
    B, nb = latents.shape[0], 2
    spec, cache = scenes._base(B, nb), {}
    fd = np.asarray(floor_dims, np.float64)
    scenes._floor(spec, cache, fd, 0.0, restitution)
    spec["shape_aux"] = np.zeros((B, nb))
    spec["igr_net"] = packed
    spec["shape_type"][:, 1] = abi.SHAPE_IGR
    spec["shape_aux"][:, 1] = 1.0
    Iunit = []
    for s in range(B):
        key = tuple(latents[s])
        if mesh_cache is None or key not in mesh_cache:
            v, f = meshsdf.igr_mesh(torch.tensor(latents[s], dtype=torch.float64), packed, res=res)
            v = v.cpu().numpy(); f = f.cpu().numpy()
            ent = (v, f, np.asarray(mass_properties.mesh_inertia(v, f, 1.0).cpu()))
            if mesh_cache is not None:
                mesh_cache[key] = ent
        else:
            ent = mesh_cache[key]
        v, f, J = ent
        spec["meshes"].append((v, f)); spec["mesh_vgrad"].append(np.zeros_like(v))
        spec["mesh_id"][s, 1] = len(spec["meshes"]) - 1
        spec["shape_prm"][s, 1, :2] = latents[s]
        spec["pose"][s, 1, 4:] = (0.0, -v[:, 1].min() + 2 * Defaults3D.EPSILON, 0.0)
        Iunit.append(J)
    T = lambda x: torch.as_tensor(x, dtype=torch.float64)
    mass, fric, force = T(mass), T(fric), T(force)
    Iu = T(np.stack(Iunit))
    one = torch.ones(B, dtype=torch.float64)
    zero = torch.zeros(B, dtype=torch.float64)
    params = dict(
        mass=torch.stack([one, mass], 1),
        inertia=torch.stack([T(spec["inertia"][:, 0]), mass[:, None, None] * Iu], 1),
        fric=torch.stack([fric, fric], 1),
        fext=torch.stack([torch.zeros(B, 6, dtype=torch.float64), torch.stack([zero, zero, zero, force[:, 0], -g * mass, force[:, 1]], 1)], 1))
    for k in ("mass", "inertia", "fric", "fext"):
        spec[k] = params[k].detach().numpy()
    w = BatchWorld3D(spec, params=params, strict_no_penetration=False, max_substeps=4 * run_steps + 64, device=device,
                     maxc=256, max_cand=8192, max_pc=128)
    return w


def fit_sysid(goal, latents, packed, target, start, run_time=1.0, max_iter=100, lr=None, conv_thresh=1e-5, res=128, log=None):
    """optim_sysid.py:184-300 for B scenes at once: `goal` in ('mass', 'force', 'friction') is estimated by gradient descent
    on sum_t |pos_t - pos_t*|^2 of the pushed body, the other two quantities are the targets'.  `target` / `start`: dicts with
    force [B,2], mass [B], fric [B] (start: only the goal's entry is read).  Learning rates as in the experiment's named
    configs (:84-100)."""
    lr = {"mass": 1e-2, "friction": 1e-3, "force": 1e-1}[goal] if lr is None else lr
    steps = int(math.ceil(run_time / Defaults3D.DT - 1e-9))
    key = {"mass": "mass", "friction": "fric", "force": "force"}[goal]
    tv = {k: torch.as_tensor(np.asarray(target[k], np.float64)) for k in ("force", "mass", "fric")}
    cache = {}
    with torch.no_grad():
        wt = push_world(latents, packed, tv["force"], tv["mass"], tv["fric"], steps, res=res, mesh_cache=cache)
        pos_t = rollout(wt, steps)[0][:, :, 1, 4:].clone()
    x = torch.tensor(np.asarray(start[key], np.float64), requires_grad=True)
    hist, last = [], None
    for e in range(max_iter):
        if x.grad is not None:
            x.grad = None
        cur = dict(tv); cur[key] = x
        w = push_world(latents, packed, cur["force"], cur["mass"], cur["fric"], steps, res=res, mesh_cache=cache)
        pos = rollout(w, steps)[0][:, :, 1, 4:]
        loss = ((pos - pos_t.to(pos)) ** 2).sum(dim=(0, 2))
        loss.sum().backward()
        l = loss.detach().cpu().numpy()
        d = (x.detach() - tv[key]).reshape(len(l), -1).norm(dim=1).numpy()
        hist.append(dict(iter=e, loss=l.copy(), value=x.detach().numpy().copy(), grad=x.grad.numpy().copy(), dist=d.copy()))
        if log:
            log("[%s] iter %3d  mean loss %.3e  mean |x - x*| %.4f  |grad| %.3e" % (goal, e, float(l.mean()), float(d.mean()), float(x.grad.abs().mean())))
        if last is not None and np.all(np.abs(last - l) < conv_thresh):
            break
        with torch.no_grad():
            x -= lr * x.grad
        last = l
    return dict(goal=goal, value=x.detach().numpy().copy(), target=tv[key].numpy().copy(), start=np.asarray(start[key], np.float64), history=hist)


def export_trajectory(path, pose, vel, **meta):
    """(T, B, nb, 13) = pose (quaternion wxyz, position) | velocity (angular, linear), as the reference's
    world.trajectory entries (world.py:376-378), one array instead of a python list per scene."""
    np.savez_compressed(path, trajectory=torch.cat([pose, vel], dim=3).detach().cpu().numpy(), **meta)


def main(argv=None):
    ap = argparse.ArgumentParser(description="batched experiment drivers (trajectory_fitting/optim_sphere, inertia_fitting/optim_shapespace and optim_primitives, system_identification/optim_sysid)")
    ap.add_argument("what", choices=["sphere", "shapespace", "inertia", "primitives", "sysid"])
    ap.add_argument("--kind", default="box", choices=sorted(PRIMITIVES))
    ap.add_argument("--goal", default="mass", choices=["mass", "force", "friction"])
    ap.add_argument("--run-time", type=float, default=1.0)
    ap.add_argument("--scenes", type=int, default=64)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args(argv)
    r = np.random.default_rng(a.seed)
    if a.what == "sphere":
        # RESULTS.md:14-47: radius error after fitting, target and start radii ~ U(0.4, 2.0) (optim_sphere.py:214-218)
        target, start = 0.4 + 1.6 * r.random(a.scenes), 0.4 + 1.6 * r.random(a.scenes)
        res = {}
        for name, kw in (("gravity, no toc", dict(use_toc_diff=False)), ("gravity, toc", dict(use_toc_diff=True)),
                         ("no gravity, toc", dict(use_toc_diff=True, use_gravity=False))):
            res[name] = fit_sphere_radius(target, start, max_iter=a.iters, log=lambda s, n=name: print("[%s] %s" % (n, s)), **kw)
        print(radius_error_table(res))
        if a.out:
            np.savez_compressed(a.out, **{k.replace(" ", "_").replace(",", ""): v["radius"] for k, v in res.items()}, target=target, start=start)
    elif a.what == "shapespace":
        from . import igr
        packed = igr.pack_weights(*scenes.geometric_init_weights(a.seed, 0.5))
        tgt, st = 0.1 * r.standard_normal((a.scenes, 2)), 0.1 * r.standard_normal((a.scenes, 2))
        res = fit_trajectory_latent(tgt, st, packed, run_time=a.run_time, max_iter=a.iters, log=print)
        print("shapespace: mean loss %.3e -> %.3e, %d scenes" % (float(res["history"][0]["loss"].mean()), float(res["history"][-1]["loss"].mean()), a.scenes))
    elif a.what == "primitives":
        # optim_primitives.py:160-175: target and start dimensions ~ U(0.5, 2.0), a random unit torque direction per scene
        n = PRIMITIVES[a.kind]
        tgt, st = 0.5 + 1.5 * r.random((a.scenes, n)), 0.5 + 1.5 * r.random((a.scenes, n))
        dirs = r.standard_normal((a.scenes, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        res = fit_inertia_primitive(a.kind, tgt, st, dirs, run_time=a.run_time if a.run_time != 1.0 else 2.0, max_iter=a.iters, log=print)
        d0, d1 = np.abs(st - tgt).mean(), np.abs(res["dims"] - tgt).mean()
        print("%s: mean |dims - target| %.4f -> %.4f, final mean loss %.3e, %d scenes" % (a.kind, d0, d1, float(res["history"][-1]["loss"].mean()), a.scenes))
    elif a.what == "sysid":
        # optim_sysid.py:184-220: a random latent, push, mass and friction per scene; the goal's start value is drawn anew
        from . import igr
        packed = igr.pack_weights(*scenes.geometric_init_weights(a.seed, 0.5))
        uni = lambda lo, hi, *sh: lo + (hi - lo) * r.random((a.scenes,) + sh)
        lat = 0.1 * r.standard_normal((a.scenes, 2))
        target = dict(force=uni(2.0, 5.0, 2), mass=uni(0.9, 1.1), fric=uni(0.01, 0.25))
        start = dict(force=uni(2.0, 5.0, 2), mass=uni(0.9, 1.1), fric=uni(0.01, 0.25))
        res = fit_sysid(a.goal, lat, packed, target, start, run_time=a.run_time, max_iter=a.iters, log=print)
        d0, d1 = np.abs(res["start"] - res["target"]).reshape(a.scenes, -1).max(1), np.abs(res["value"] - res["target"]).reshape(a.scenes, -1).max(1)
        print("%s: |x - x*| start mean %.4f -> final mean %.4f (max %.4f), %d scenes" % (a.goal, d0.mean(), d1.mean(), d1.max(), a.scenes))
    else:
        from . import igr
        packed = igr.pack_weights(*scenes.geometric_init_weights(a.seed, 0.5))
        tgt, st = 0.1 * r.standard_normal((a.scenes, 2)), 0.1 * r.standard_normal((a.scenes, 2))
        dirs = r.standard_normal((a.scenes, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        fit_inertia_latent(tgt, st, dirs, packed, max_iter=a.iters, log=print)


if __name__ == "__main__":
    main()
