"""Golden for the trajectory-fitting loss (TEST INFRA, build container only):  python -m oracle.gen.gen_trajloss_golden

The reference's own `make_world`, `run_world_fixed_dt(world, run_time, detach_2nd_bounce)` and `trajectory_loss(world,
world_target)` (experiments/trajectory_fitting/optim_sphere.py:77-177), imported as they are, on the wall + floor bounce scene:
a target sphere and a start sphere of another radius, analytic meshes / inertias.  The sphere hits the wall inside a step, so
dt is halved and the trajectory gets extra entries around the contact -- `trajectory_loss` sums over EVERY accepted sub-step
(entries are stamped with the time at the START of their sub-step, world.py:373-379) and divides by their number.
Stored: both trajectories (t, pose, vel per entry), the loss and d loss / d radius, with and without detach_2nd_bounce.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refshim  # noqa: E402

refshim.install()
import optim_sphere as O  # noqa: E402   (the reference's script; sacred is a stand-in whose decorators return the function)

OUT = os.path.join(ROOT, "tests", "golden")
KW = dict(use_toc_diff=True, use_friction=True, use_wall=True, use_floor=True, use_gravity=True, custom_mesh=True, custom_inertia=True)


def traj(world):
    nb = len(world.bodies)
    return (np.array([float(e[0]) for e in world.trajectory]), np.stack([e[1].detach().numpy().reshape(nb, 7) for e in world.trajectory]),
            np.stack([e[2].detach().numpy().reshape(nb, 6) for e in world.trajectory]))


def main():
    d = {}
    run_time = 1.5
    r_target, r_start = 0.7, 0.9
    d["r_target"], d["r_start"], d["run_time"] = r_target, r_start, run_time
    wt, _ = O.make_world(torch.tensor(r_target, dtype=torch.double), **KW)
    O.run_world_fixed_dt(wt, run_time)
    d["target_t"], d["target_p"], d["target_v"] = traj(wt)
    for tag, detach in (("plain", False), ("detach", True)):
        rad = torch.tensor(r_start, dtype=torch.double, requires_grad=True)
        w, _ = O.make_world(rad, **KW)
        O.run_world_fixed_dt(w, run_time, detach_2nd_bounce=detach)
        loss = O.trajectory_loss(w, wt)
        g, = torch.autograd.grad(loss, rad)
        d[tag + "_t"], d[tag + "_p"], d[tag + "_v"] = traj(w)
        d[tag + "_loss"], d[tag + "_grad"] = float(loss), float(g)
        print(tag, "entries", len(w.trajectory), "target entries", len(wt.trajectory), "loss", float(loss), "d loss / d rad", float(g))
    np.savez_compressed(os.path.join(OUT, "trajectory_loss_bounce.npz"), **d)


if __name__ == "__main__":
    main()
