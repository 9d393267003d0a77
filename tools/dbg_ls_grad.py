"""Dev aid (GPU box): level-set sphere / cylinder through the World3D class API, gradient vs the reference golden."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch, rollout_helpers as R
from diffsdfsim_amd.physics3d import Gravity3D, SDFBox, SDFCylinder, SDFSphere, TotalConstraint3D, World3D
for name in ("tmp_lssphere_grad", "tmp_lscyl_grad"):
    g = R.load_rollout(name)
    floor = SDFBox([0, -0.5, 0], [4.0, 1.0, 4.0], custom_mesh=True, custom_inertia=True, restitution=0.3, fric_coeff=0.4)
    if "sphere" in name:
        prm = [torch.tensor(0.3, dtype=torch.double, requires_grad=True)]
        b = SDFSphere([0.0, 0.45, 0.0], prm[0], vel=[0, 0, 1.0, 0.5, -0.5, 0], restitution=0.3, fric_coeff=0.4, custom_mesh=False, custom_inertia=False)
        n = 12
    else:
        prm = [torch.tensor(0.25, dtype=torch.double, requires_grad=True), torch.tensor(0.6, dtype=torch.double, requires_grad=True)]
        b = SDFCylinder([0.0, 0.2505, 0.0], prm[0], prm[1], vel=[0, 0, 1.0, 0.5, 0, 0], restitution=0.1, fric_coeff=0.3, custom_mesh=False, custom_inertia=False)
        n = 8
    b.add_force(Gravity3D())
    assert (len(b.verts), len(b.faces)) == tuple(g["meshsize_1"]), ((len(b.verts), len(b.faces)), g["meshsize_1"])
    w = World3D([floor, b], [TotalConstraint3D(floor)])
    for _ in range(n):
        w.step(fixed_dt=True)
    loss = (floor.p[4:] ** 2).sum() + (b.p[4:] ** 2).sum()
    loss.backward()
    print(name, "pose dev %.1e" % np.abs(b.p.detach().cpu().numpy() - g["traj_p"][-1][1]).max(),
          "grads", [float(p.grad) for p in prm], "ref", [float(g["grad_%d" % i]) for i in range(len(prm))])
