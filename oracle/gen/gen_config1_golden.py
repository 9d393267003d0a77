"""Config 1 (BASELINE.json configs[0]): the reference's own 2-D CPU case -- Circle bouncing on a Rect,
lcp_physics.physics World + DiffContactHandler (analytic contacts), 50 steps fwd+bwd.

The 2-D host logic (bodies, GJK contacts) is outside the hot path ("config 1 only");
what config 1 exercises on the path is boundary B1.  This script records every LCPFunction call the
reference makes during the rollout (operands, solution, multipliers, and the upstream gradient autograd
feeds into its backward) -> tests/golden/config1_lcp.npz, plus the final state and d loss / d rad.
Run in the build container only:  python -m oracle.gen.gen_config1_golden
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refshim  # noqa: E402

refshim.install()
import lcp_physics.lcp.solvers.batch as B  # noqa: E402
import lcp_physics.physics.engines as engines  # noqa: E402
from lcp_physics.physics.bodies import Circle, Rect  # noqa: E402
from lcp_physics.physics.constraints import TotalConstraint  # noqa: E402
from lcp_physics.physics.forces import Gravity  # noqa: E402
from lcp_physics.physics.world import World  # noqa: E402


def main():
    rad = torch.tensor(20.0, dtype=torch.double, requires_grad=True)
    floor = Rect([500, 600], [1000, 50], restitution=0.5, fric_coeff=0.9)
    ball = Circle([500, 480], rad, vel=[0, 30, 0], restitution=0.5, fric_coeff=0.9)
    ball.add_force(Gravity(g=100))
    calls = []
    orig_fn, orig_fwd = engines.LCPFunction, B.forward
    state = {}

    def fwd(*a, **k):
        out = orig_fwd(*a, **k)
        state["xyzs"] = [None if o is None else o.clone() for o in out]
        return out

    def spy(**kw):
        fn = orig_fn(**kw)

        def call(*ops):
            z = fn(*ops)
            rec = {"kw": kw, "ops": [o.detach().clone() for o in ops], "z": z.detach().clone(),
                   "lam": state["xyzs"][2], "slack": state["xyzs"][3], "nu": state["xyzs"][1]}
            z.register_hook(lambda g, rec=rec: rec.__setitem__("dl", g.clone()))
            calls.append(rec)
            return z
        return call

    B.forward = fwd
    w = World([floor, ball], [TotalConstraint(floor)], dt=1.0 / 30)
    w.engine.lcp_solver = spy
    for _ in range(50):
        w.step()
    loss = (ball.pos ** 2).sum()
    loss.backward()
    B.forward = orig_fwd
    d = {"n_calls": np.int64(len(calls)), "final_p": torch.cat([b.p for b in (floor, ball)]).detach().numpy(),
         "loss": float(loss), "drad": rad.grad.numpy(), "t_final": float(w.t), "n_substeps": np.int64(len(w.trajectory))}
    for i, c in enumerate(calls):
        for n, o in zip("QpGhAbF", c["ops"]):
            d["c%d_%s" % (i, n)] = o.numpy()
        d["c%d_z" % i] = c["z"].numpy(); d["c%d_lam" % i] = c["lam"].numpy(); d["c%d_slack" % i] = c["slack"].numpy()
        d["c%d_nu" % i] = c["nu"].numpy(); d["c%d_dl" % i] = c["dl"].numpy(); d["c%d_max_iter" % i] = np.int64(c["kw"]["max_iter"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "config1_lcp.npz"), **d)
    print("config 1: %d LCP calls in %d sub-steps, final p %s, dL/drad %.6f" % (len(calls), len(w.trajectory), d["final_p"], float(rad.grad)),
          [tuple(c["ops"][2].shape) for c in calls][:5])


def polygons():
    """A second 2-D scene with polygon / polygon and circle / polygon contacts in one world (a tilted box landing on a corner
    and tipping onto its face, a ball rolling into it, on a pinned slab): contact pairs per step, final poses and
    d loss / d (box width, ball radius) -> tests/golden/config1_polygons.npz."""
    wd = torch.tensor(80.0, dtype=torch.double, requires_grad=True)
    rad = torch.tensor(25.0, dtype=torch.double, requires_grad=True)
    floor = Rect([500, 600], [1000, 50], restitution=0.2, fric_coeff=0.6)
    box = Rect([0.3, 420, 520], torch.stack([wd, wd.new_tensor(50.0)]), restitution=0.2, fric_coeff=0.6)
    ball = Circle([560, 500], rad, vel=[0, -150, 0], restitution=0.2, fric_coeff=0.6)
    for b in (box, ball):
        b.add_force(Gravity(g=100))
    w = World([floor, box, ball], [TotalConstraint(floor)], dt=1.0 / 30)
    pairs, traj = [], []
    for _ in range(60):
        w.step()
        row = np.full((8, 2), -1, np.int64)
        for k, c in enumerate(w.contacts):
            row[k] = (c[1], c[2])
        pairs.append(row)
        traj.append(torch.cat([b.p for b in (floor, box, ball)]).detach().numpy())
    loss = (box.p ** 2).sum() + (ball.pos ** 2).sum()
    gw, gr = torch.autograd.grad(loss, [wd, rad])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "config1_polygons.npz"), pairs=np.stack(pairs), traj_p=np.stack(traj),
                        loss=float(loss), g_width=float(gw), g_rad=float(gr), n_substeps=np.int64(len(w.trajectory)), t_final=float(w.t))
    print("polygon scene: %d sub-steps, loss %.6f, d/d width %.6f, d/d rad %.6f, pairs seen %s" %
          (len(w.trajectory), float(loss), float(gw), float(gr), sorted({tuple(r) for s_ in pairs for r in s_ if r[0] >= 0})))


def hulls():
    """General convex polygons (`Hull`: centroid shift of the vertices, polygon inertia; bodies.py:196-254): a pentagon and a
    triangle dropped on the slab next to each other, the pentagon given a spin and scaled by a differentiable factor
    -> tests/golden/config1_hulls.npz."""
    from lcp_physics.physics.bodies import Hull
    sc = torch.tensor(1.0, dtype=torch.double, requires_grad=True)
    pent = [[40.0, 0.0], [12.0, 38.0], [-32.0, 24.0], [-32.0, -24.0], [12.0, -38.0]]
    tri = [[30.0, 20.0], [-30.0, 20.0], [0.0, -35.0]]
    floor = Rect([500, 600], [1000, 50], restitution=0.3, fric_coeff=0.5)
    a = Hull([430, 500], [sc * torch.tensor(v, dtype=torch.double) for v in pent], vel=[1.5, 20, 0], restitution=0.3, fric_coeff=0.5)
    b = Hull([520, 520], [torch.tensor(v, dtype=torch.double) for v in tri], vel=[0, -30, 0], restitution=0.3, fric_coeff=0.5)
    for x in (a, b):
        x.add_force(Gravity(g=100))
    w = World([floor, a, b], [TotalConstraint(floor)], dt=1.0 / 30)
    pairs, traj = [], []
    for _ in range(50):
        w.step()
        row = np.full((8, 2), -1, np.int64)
        for k, c in enumerate(w.contacts):
            row[k] = (c[1], c[2])
        pairs.append(row)
        traj.append(torch.cat([x.p for x in (floor, a, b)]).detach().numpy())
    loss = (a.p ** 2).sum() + (b.p ** 2).sum()
    gs, = torch.autograd.grad(loss, [sc])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "config1_hulls.npz"), pairs=np.stack(pairs), traj_p=np.stack(traj),
                        loss=float(loss), g_scale=float(gs), n_substeps=np.int64(len(w.trajectory)),
                        inertia=np.array([float(a.ang_inertia), float(b.ang_inertia)]))
    print("hull scene: %d sub-steps, loss %.6f, d/d scale %.6f, inertias %.4f %.4f, max contacts %d" %
          (len(w.trajectory), float(loss), float(gs), float(a.ang_inertia), float(b.ang_inertia), max((r[:, 0] >= 0).sum() for r in pairs)))


if __name__ == "__main__":
    main()
    polygons()
    hulls()
