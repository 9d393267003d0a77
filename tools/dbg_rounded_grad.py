"""Dev aid (GPU box): partial gradients of the rounded-box drop (SDF only / + mesh / + inertia)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffsdfsim_amd.physics3d import SDFBox, SDFBoxRounded, World3D
from diffsdfsim_amd.physics3d.constraints import TotalConstraint3D
from diffsdfsim_amd.physics3d.forces import Gravity3D

for mode in ("sdf_only", "sdf_mesh", "all"):
    dims = torch.tensor([0.6, 0.5, 0.7], dtype=torch.double, requires_grad=True)
    floor = SDFBox([0, -0.5, 0], [4.0, 1.0, 4.0], custom_mesh=True, custom_inertia=True, restitution=0.3, fric_coeff=0.4)
    b = SDFBoxRounded(torch.tensor([0.25, 0.1, -0.2, 0.0, 0.5, 0.0], dtype=torch.double), dims, 0.15,
                      vel=torch.tensor([0.3, -0.1, 0.2, 0.6, -0.4, 0.1], dtype=torch.double), restitution=0.3, fric_coeff=0.4)
    b.add_force(Gravity3D())
    if mode != "all":
        b.ang_inertia = b.ang_inertia.detach()
    if mode == "sdf_only":
        b.verts_t = b.verts_t.detach()
    w = World3D([floor, b], [TotalConstraint3D(floor)])
    for _ in range(10):
        w.step(fixed_dt=True)
    loss = (floor.p[4:] ** 2).sum() + (b.p[4:] ** 2).sum()
    loss.backward()
    print(mode, dims.grad.numpy())
