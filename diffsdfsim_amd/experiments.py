"""Batched experiment drivers: the trajectory-fitting loop of
`experiments/trajectory_fitting/optim_sphere.py:114-177` (make a world from the current parameter estimate, roll it out
with fixed outer steps, compare with a target trajectory, step the parameter along the gradient) for a whole batch of
independent scenes at once, without sacred / pyrender.

    rollout(world, steps)                 -> poses [T,B,nb,7], velocities [T,B,nb,6]   (autograd-connected)
    trajectory_loss(traj, traj_target)    -> [B]  mean over time of |pos - pos_target|^2 of the last body (optim_sphere.py:114-160)
    sphere_world(radii, ...)              -> BatchWorld3D of config 2's scene with the given per-scene radii; shape and
                                             inertia are functions of `radii`, so d loss / d radii flows
    fit_sphere_radius(...)                -> the optimisation loop; returns the history
    export_trajectory(path, ...)          -> .npz of (T, B, nb, 13) trajectories (pyrender-free export)

    python -m diffsdfsim_amd.experiments --scenes 64 --steps 30 --iters 15
"""
import argparse

import numpy as np
import torch

from . import meshes, scenes
from . import world_abi as abi
from .physics3d import BatchWorld3D


def rollout(world, steps):
    P, V = [], []
    for _ in range(steps):
        world.step()
        P.append(world.pose)
        V.append(world.vel)
    return torch.stack(P), torch.stack(V)


def trajectory_loss(traj_pose, target_pose, body=-1):
    """Fixed outer steps put both trajectories on the same time grid, so the reference's nearest-time search
    (optim_sphere.py:121-139) is the identity pairing."""
    d = traj_pose[:, :, body, 4:] - target_pose[:, :, body, 4:].to(traj_pose)
    return (d * d).sum(dim=2).mean(dim=0)


def sphere_world(radii, y0, vx, floor_dims=(20.0, 1.0, 20.0), mu=0.25, rest=0.5, g=10.0, steps=64, toc=True, device=None):
    """Config 2's scene for explicit per-scene radius / drop height / lateral speed."""
    rad = np.asarray(radii.detach().cpu() if torch.is_tensor(radii) else radii, np.float64)
    B = len(rad)
    spec, cache = scenes._base(B, 2), {}
    scenes._floor(spec, cache, floor_dims, mu, rest)
    uv, uf = meshes.icosphere(4)
    for s in range(B):
        spec["mesh_id"][s, 1] = scenes._add_mesh(spec, cache, ("sphere", float(rad[s])), lambda r=float(rad[s]): (uv * r, uf, uv.copy()))
        spec["shape_type"][s, 1] = abi.SHAPE_SPHERE
        spec["shape_prm"][s, 1, 0] = rad[s]
        spec["pose"][s, 1, 4:] = (0.0, y0[s], 0.0)
        spec["vel"][s, 1, 3] = vx[s]
        spec["inertia"][s, 1] = 0.4 * rad[s] ** 2 * np.eye(3)
        spec["fric"][s, 1] = mu
        spec["restitution"][s, 1] = rest
        spec["fext"][s, 1, 4] = -g
    params = {}
    if torch.is_tensor(radii) and radii.requires_grad:
        r = radii.to(torch.float64)
        prm = torch.tensor(spec["shape_prm"], dtype=torch.float64)
        prm = torch.cat([prm[:, :1], torch.stack([r, torch.zeros_like(r), torch.zeros_like(r)], 1)[:, None].to(prm)], 1)
        inertia = torch.tensor(spec["inertia"], dtype=torch.float64)
        ball = (0.4 * r * r)[:, None, None].to(inertia) * torch.eye(3, dtype=torch.float64)      # 2/5 m r^2 (bodies.py:993-994)
        inertia = torch.cat([inertia[:, :1], ball[:, None]], 1)
        params = dict(shape_prm=prm, inertia=inertia)
    return BatchWorld3D(spec, params=params, time_of_contact_diff=toc, max_substeps=4 * steps + 64, device=device)


def fit_sphere_radius(target_radii, init_radii, y0, vx, steps=30, iters=20, lr=0.05, log=None):
    """Adam on the radius of every scene at once (optim_sphere.py:163-250 runs one scene per process; its per-parameter
    step normalisation keeps the scenes of the batch independent of each other).  A new world is built from the current
    estimate in every iteration, like the reference's make_world."""
    with torch.no_grad():
        target_pose, _ = rollout(sphere_world(torch.as_tensor(target_radii), y0, vx, steps=steps), steps)
    rad = torch.tensor(np.asarray(init_radii, np.float64), requires_grad=True)
    opt = torch.optim.Adam([rad], lr=lr)
    hist = []
    for it in range(iters):
        opt.zero_grad()
        world = sphere_world(rad, y0, vx, steps=steps)
        pose, _ = rollout(world, steps)
        loss = trajectory_loss(pose, target_pose)
        loss.sum().backward()
        hist.append(dict(iter=it, loss=loss.detach().cpu().numpy().copy(), radius=rad.detach().numpy().copy(),
                         grad=rad.grad.numpy().copy()))
        if log:
            log("iter %2d  mean loss %.3e  mean |r - r*| %.4f" % (it, float(loss.detach().mean()), float((rad.detach() - torch.as_tensor(target_radii)).abs().mean())))
        opt.step()
        with torch.no_grad():
            rad.clamp_(0.2, 0.9)
    return hist, target_pose


def export_trajectory(path, pose, vel, **meta):
    """(T, B, nb, 13) = pose (quaternion wxyz, position) | velocity (angular, linear), as the reference's
    world.trajectory entries (world.py:376-378), one array instead of a python list per scene."""
    np.savez_compressed(path, trajectory=torch.cat([pose, vel], dim=3).detach().cpu().numpy(), **meta)


def main(argv=None):
    ap = argparse.ArgumentParser(description="batched sphere-radius fitting (trajectory_fitting/optim_sphere, all scenes at once)")
    ap.add_argument("--scenes", type=int, default=64)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--iters", type=int, default=15)
    ap.add_argument("--lr", type=float, default=0.03)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args(argv)
    r = np.random.default_rng(a.seed)
    target = 0.4 + 0.2 * r.random(a.scenes)
    init = target + 0.08 * (r.random(a.scenes) - 0.5)
    y0, vx = 0.7 + 0.5 * r.random(a.scenes), r.random(a.scenes)
    hist, target_pose = fit_sphere_radius(target, init, y0, vx, steps=a.steps, iters=a.iters, lr=a.lr, log=print)
    if a.out:
        np.savez_compressed(a.out, target=target, init=init, radius=np.stack([h["radius"] for h in hist]), loss=np.stack([h["loss"] for h in hist]))


if __name__ == "__main__":
    main()
