"""Golden vector for the contact-free branch (`optim_shapespace.make_world`,
experiments/.../optim_shapespace.py:71-92 with a primitive in place of the IGR body): one body, translation locked by
X/Y/ZConstraint, a torque along a fixed unit direction for t < 0.3, no contacts -- the engine takes the linear-solve
branch (`lcp_physics/physics/engines.py:40-54`).  Stores every step's (t, pose, velocity) and
d sum(omega_T^2) / d dims (through the analytic inertia).

Run in the build container only:  python -m oracle.gen.gen_config5_golden
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refshim  # noqa: E402

refshim.install()
from sdf_physics.physics3d.bodies import SDFBox  # noqa: E402
from sdf_physics.physics3d.constraints import XConstraint, YConstraint, ZConstraint  # noqa: E402
from sdf_physics.physics3d.forces import ExternalForce3D  # noqa: E402
from sdf_physics.physics3d.world import World3D  # noqa: E402

DIR = np.array([0.6, -0.3, 0.74]); DIR = DIR / np.linalg.norm(DIR)
NSTEPS, T_OFF, MAG = 18, 0.3, 0.8


def main():
    dims = torch.tensor([1.0, 0.6, 0.4], dtype=torch.double, requires_grad=True)
    body = SDFBox([0, 0, 0], dims, mass=1.5, custom_mesh=True, custom_inertia=True)
    tq = torch.tensor(np.concatenate([DIR, np.zeros(3)]))

    def force(t):
        return tq if t < T_OFF else ExternalForce3D.ZEROS
    body.add_force(ExternalForce3D(force, multiplier=MAG))
    w = World3D([body], [XConstraint(body), YConstraint(body), ZConstraint(body)])
    for _ in range(NSTEPS):
        w.step(fixed_dt=True)
    loss = (body.v[:3] ** 2).sum()
    (g,) = torch.autograd.grad(loss, [dims])
    d = dict(dims=dims.detach().numpy(), mass=1.5, dir=DIR, t_off=T_OFF, mag=MAG, nsteps=NSTEPS, dt=w.dt,
             traj_t=np.array([float(e[0]) for e in w.trajectory]),
             traj_p=np.stack([e[1].detach().numpy() for e in w.trajectory]),
             traj_v=np.stack([e[2].detach().numpy() for e in w.trajectory]), loss=float(loss), grad_dims=g.numpy())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "config5_spin.npz"), **d)
    print("config5_spin: steps", len(d["traj_t"]), "omega_T", d["traj_v"][-1][:3], "loss", float(loss), "grad", g.numpy())


if __name__ == "__main__":
    main()
