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
from sdf_physics.physics3d.bodies import (SDFBowl, SDFBox, SDFBoxRounded, SDFBrick, SDFCylinder, SDFSphere,  # noqa: E402
                                          get_ang_inertia)

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
    np.savez_compressed(os.path.join(OUT, "sdf_query.npz"), **out)
    np.savez_compressed(os.path.join(OUT, "mesh_inertia.npz"), **inert)


if __name__ == "__main__":
    main()
