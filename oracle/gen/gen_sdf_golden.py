"""Generate G4 golden vectors: primitive SDF queries and mesh inertia from the reference's CPU path.

Run in the build container only:  python -m oracle.gen.gen_sdf_golden
Writes tests/golden/sdf_query.npz:  for SDFBox / SDFSphere / SDFCylinder / SDFBoxRounded / SDFBrick / SDFBowl
(`sdf_physics/physics3d/bodies.py:778-1027`; the level-set meshes of the last three come from the functional marching-cubes
stand-in, which the queries do not depend on)
the outputs of `SDF3D.query_sdfs(pts, return_grads=True, return_overlapmask=True)` (`bodies.py:721-760`) on a fixed
point set with random, surface, edge / corner / axis, centre and outside-the-query-cube points; and
tests/golden/mesh_inertia.npz: `get_ang_inertia(verts, faces, mass)` (`bodies.py:260-395`) on the custom meshes.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refshim  # noqa: E402

refshim.install()
from sdf_physics.physics3d.bodies import (SDFBowl, SDFBox, SDFBoxRounded, SDFBrick, SDFCylinder, SDFGrid3D,  # noqa: E402
                                          SDFSphere, get_ang_inertia)

OUT = os.path.join(ROOT, "tests", "golden")


def points(scale, dims, seed):
    r = np.random.default_rng(seed)
    hd = np.asarray(dims, np.float64) / 2
    pts = [r.uniform(-1.2 * scale, 1.2 * scale, (160, 3)),                 # anywhere, some outside the query cube
           r.uniform(-1, 1, (60, 3)) * hd,                                  # inside
           np.array([[sx * hd[0], sy * hd[1], sz * hd[2]] for sx in (-1, 0, 1) for sy in (-1, 0, 1) for sz in (-1, 0, 1)]),
           np.array([[0.0, 0.0, 0.0], [scale, 0, 0], [0, -scale, 0], [scale, scale, scale], [1.0001 * scale, 0, 0]]),
           np.array([[hd[0] + 0.1, hd[1] + 0.1, 0.0], [hd[0] + 0.1, hd[1] + 0.1, hd[2] + 0.1], [0.0, 0.0, hd[2] + 0.05],
                     [0.3 * hd[0], 0.3 * hd[0], 0.0], [-0.2 * hd[0], 0.2 * hd[0], -0.2 * hd[0]]])]
    return np.concatenate(pts, 0)


def main():
    out, inert = {}, {}
    # (name, reference body, bounding dims for the point set, parameters in world units as the C ABI takes them)
    cases = [("box", SDFBox([0, 0, 0], torch.tensor([0.9, 1.1, 1.3], dtype=torch.double), custom_mesh=True, custom_inertia=True),
              [0.9, 1.1, 1.3], [0.9, 1.1, 1.3]),
             ("sphere", SDFSphere([0, 0, 0], torch.tensor(0.55, dtype=torch.double), custom_mesh=True, custom_inertia=True),
              [1.1, 1.1, 1.1], [0.55, 0.0, 0.0]),
             ("cylinder", SDFCylinder([0, 0, 0], torch.tensor(0.4, dtype=torch.double), torch.tensor(1.2, dtype=torch.double),
                                      custom_mesh=True, custom_inertia=True), [0.8, 0.8, 1.2], [0.4, 1.2, 0.0]),
             # fourth parameter = the corner radius r (a constant of the body)
             ("rounded", SDFBoxRounded([0, 0, 0], torch.tensor([0.9, 1.1, 1.3], dtype=torch.double), 0.2), [0.9, 1.1, 1.3],
              [0.9, 1.1, 1.3, 0.2]),
             ("brick", SDFBrick([0, 0, 0], torch.tensor([1.0, 0.8, 0.6], dtype=torch.double), 0.15), [1.0, 0.8, 0.6],
              [1.0, 0.8, 0.6, 0.15]),
             ("bowl", SDFBowl([0, 0, 0], torch.tensor(0.8, dtype=torch.double), torch.tensor(0.1, dtype=torch.double),
                              custom_mesh=True), [1.8, 1.8, 1.8], [0.8, 0.1, 0.0])]
    for k, (name, body, dims, prm) in enumerate(cases):
        scale = float(body.scale)
        pts = points(scale, dims, 10 + k)
        sdf, grad, mask = body.query_sdfs(torch.tensor(pts), return_grads=True, return_overlapmask=True)
        out[name + "_pts"] = pts
        out[name + "_prm"] = np.asarray(prm, np.float64)
        out[name + "_scale"] = scale
        out[name + "_sdf"] = sdf.detach().numpy()
        out[name + "_grad"] = grad.detach().numpy()
        out[name + "_mask"] = mask.numpy()
        if name in ("rounded", "brick"):
            continue          # level-set meshes: order is the stand-in's, not a reference fact
        v, f = body.verts.detach(), body.faces
        J = get_ang_inertia(v, f, torch.tensor(2.5, dtype=torch.double))
        inert[name + "_verts"] = v.numpy(); inert[name + "_faces"] = f.numpy().astype(np.int32)
        inert[name + "_J"] = J.numpy(); inert[name + "_mass"] = 2.5
        print(name, "scale", scale, "n", len(pts), "inside cube", int(mask.sum()), "J diag", np.diag(J.numpy()))
    # SDFGrid3D (bodies.py:763-775): an ellipsoid-like level set sampled on 24^3.  grid_interp comes from the stand-in
    # (plain trilinear interpolation), so this vector pins the index mapping, the central-difference field, the masks and
    # the normalisations of the reference's grid_sdf / grid_sdf_grad / query_sdfs around it.
    n = 24
    lin = np.linspace(-1.0, 1.0, n)
    X, Y, Z = np.meshgrid(lin, lin, lin, indexing="ij")
    grid = np.sqrt((X / 0.8) ** 2 + (Y / 0.6) ** 2 + (Z / 0.7) ** 2) - 1.0 + 0.05 * np.sin(3 * X) * np.cos(2 * Y)
    body = SDFGrid3D([0, 0, 0], 0.9, torch.tensor(grid))
    r = np.random.default_rng(77)
    pts = np.concatenate([r.uniform(-1.1 * 0.9, 1.1 * 0.9, (200, 3)), np.array([[0.9, 0.9, 0.9], [-0.9, 0.0, 0.0], [0.0, 0.0, 0.0]]),
                          0.9 * np.stack([lin[[0, 5, 23, 11]], lin[[3, 0, 23, 12]], lin[[7, 23, 0, 12]]], 1)])
    sdf, grad, mask = body.query_sdfs(torch.tensor(pts), return_grads=True, return_overlapmask=True)
    out.update(grid_grid=grid, grid_scale=0.9, grid_pts=pts, grid_sdf=sdf.detach().numpy(), grid_grad=grad.detach().numpy(),
               grid_mask=mask.numpy())
    print("grid", "inside cube", int(mask.sum()), "mesh", tuple(body.verts.shape), tuple(body.faces.shape))
    np.savez_compressed(os.path.join(OUT, "sdf_query.npz"), **out)
    np.savez_compressed(os.path.join(OUT, "mesh_inertia.npz"), **inert)


if __name__ == "__main__":
    main()
