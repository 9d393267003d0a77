"""CPU, build container only: the drop-in boundaries B1 (LCPFunction), B2 (engine plug-in) and B2' (contact handler)
exercised by the REFERENCE's own code.

BASELINE.json configs[0] is the reference's CPU case: lcp_physics' 2-D world (Circle bouncing on a Rect, analytic
contacts, `lcp_physics/physics/world.py`, `engines.py:31-83`), 50 steps forward + backward.  Here that world runs
unmodified, with one substitution: `engine.lcp_solver`, the `LCPFunction` it instantiates at engines.py:81, is replaced by
an autograd Function whose forward / backward are this build's dense LCP kernels (csrc/lcp_dense.hip) -- compiled for the
test-only CPU emulator, since the reference and a GPU never exist on the same machine.  Final state, loss and
d loss / d radius must equal what the reference produced with its own solver (tests/golden/config1_lcp.npz).
Skipped where /root/reference is absent (the GPU box); the same kernels run there through
test_lcp_dense_gpu.py::test_config1_reference_cpu_case_through_lcpfunction on the recorded operands."""
import os

import numpy as np
import pytest
import torch

from emu import emu
from helpers import GOLDEN

REF = os.environ.get("DIFFSDFSIM_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def kernel_lcp_function(calls):
    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, Q, p, G, h, A, b, F, max_iter):
            neq = A.shape[1] if A.numel() else 0
            ops = [t.detach().numpy() for t in (Q, p, G, h)] + [A.detach().numpy() if neq else np.zeros((1, 0, Q.shape[1])),
                                                                  b.detach().numpy() if neq else np.zeros((1, 0)), F.detach().numpy()]
            z, lam, s, nu, it, st = emu.lcp_dense_forward(*ops, max_iter=max_iter)
            ctx.ops, ctx.sol, ctx.neq = ops, (z, lam, s, nu), neq
            calls.append(int(it[0]))
            return torch.as_tensor(z)

        @staticmethod
        def backward(ctx, dl):
            Q, p, G, h, A, b, F = ctx.ops
            z, lam, s, nu = ctx.sol
            dQ, dp, dG, dh, dA, db, dF = emu.lcp_dense_backward(Q, G, A, F, z, lam, s, nu, dl.numpy())
            t = torch.as_tensor
            return t(dQ), t(dp), t(dG), t(dh), (t(dA) if ctx.neq else None), (t(db) if ctx.neq else None), t(dF), None

    def LCPFunction(max_iter=20, **_kw):      # the constructor signature of lcp.py:43-46, as engines.py:81 calls it
        return lambda Q, p, G, h, A, b, F: Fn.apply(Q, p, G, h, A, b, F, max_iter)
    return LCPFunction


def test_reference_2d_world_on_the_builds_lcp_kernels():
    from oracle import refshim
    refshim.install()
    from lcp_physics.physics.bodies import Circle, Rect
    from lcp_physics.physics.constraints import TotalConstraint
    from lcp_physics.physics.forces import Gravity
    from lcp_physics.physics.world import World

    g = np.load(os.path.join(GOLDEN, "config1_lcp.npz"))
    rad = torch.tensor(20.0, dtype=torch.double, requires_grad=True)
    floor = Rect([500, 600], [1000, 50], restitution=0.5, fric_coeff=0.9)
    ball = Circle([500, 480], rad, vel=[0, 30, 0], restitution=0.5, fric_coeff=0.9)
    ball.add_force(Gravity(g=100))
    calls = []
    w = World([floor, ball], [TotalConstraint(floor)], dt=1.0 / 30)
    w.engine.lcp_solver = kernel_lcp_function(calls)
    for _ in range(50):
        w.step()
    loss = (ball.pos ** 2).sum()
    loss.backward()
    assert len(calls) == int(g["n_calls"]) and len(w.trajectory) == int(g["n_substeps"])
    final_p = torch.cat([b.p for b in (floor, ball)]).detach().numpy()
    assert np.abs(final_p - g["final_p"]).max() < 1e-8 * np.abs(g["final_p"]).max()
    assert abs(float(loss) - float(g["loss"])) < 1e-8 * abs(float(g["loss"]))
    assert abs(float(rad.grad) - float(g["drad"])) < 1e-6 * abs(float(g["drad"])), (float(rad.grad), float(g["drad"]))


def emu_contact_handler():
    """The product's handler class (diffsdfsim_amd/physics2d) over the emulator build of csrc/contacts2d.hip."""
    from diffsdfsim_amd.physics2d import make_handler
    t = torch.as_tensor

    def fwd(kind, nv, pos, rad, verts, sat_in, eps):
        out, count, sat_out = emu.contacts2d_forward(kind.numpy(), nv.numpy(), pos.detach().numpy(), rad.detach().numpy(),
                                                     verts.detach().numpy(), sat_in.numpy(), eps)
        return t(out), t(count), t(sat_out)

    def bwd(kind, nv, pos, rad, verts, sat_in, eps, gout):
        return tuple(t(x) for x in emu.contacts2d_backward(kind.numpy(), nv.numpy(), pos.numpy(), rad.numpy(), verts.numpy(),
                                                           sat_in.numpy(), eps, gout.numpy()))
    return make_handler((fwd, bwd))


def test_reference_2d_world_on_the_builds_contact_and_lcp_kernels():
    """Config 1 with BOTH of its path functions replaced: the analytic contact handler (R18, the `contact_callback=` seam,
    lcp_physics/physics/world.py:44-52) and the LCP (B1).  Same golden, same tolerances."""
    from oracle import refshim
    refshim.install()
    from lcp_physics.physics.bodies import Circle, Rect
    from lcp_physics.physics.constraints import TotalConstraint
    from lcp_physics.physics.forces import Gravity
    from lcp_physics.physics.world import World

    g = np.load(os.path.join(GOLDEN, "config1_lcp.npz"))
    rad = torch.tensor(20.0, dtype=torch.double, requires_grad=True)
    floor = Rect([500, 600], [1000, 50], restitution=0.5, fric_coeff=0.9)
    ball = Circle([500, 480], rad, vel=[0, 30, 0], restitution=0.5, fric_coeff=0.9)
    ball.add_force(Gravity(g=100))
    calls = []
    w = World([floor, ball], [TotalConstraint(floor)], dt=1.0 / 30, contact_callback=emu_contact_handler())
    w.engine.lcp_solver = kernel_lcp_function(calls)
    for _ in range(50):
        w.step()
    loss = (ball.pos ** 2).sum()
    loss.backward()
    assert len(calls) == int(g["n_calls"]) and len(w.trajectory) == int(g["n_substeps"])
    final_p = torch.cat([b.p for b in (floor, ball)]).detach().numpy()
    assert np.abs(final_p - g["final_p"]).max() < 1e-8 * np.abs(g["final_p"]).max()
    assert abs(float(loss) - float(g["loss"])) < 1e-8 * abs(float(g["loss"]))
    assert abs(float(rad.grad) - float(g["drad"])) < 1e-6 * abs(float(g["drad"])), (float(rad.grad), float(g["drad"]))


def test_reference_2d_world_polygon_scene_with_the_builds_contact_handler():
    """Polygon against polygon and circle against polygon in one world (a tilted box and a ball dropped on a floor slab, the
    box landing on a corner and tipping onto its face, the ball rolling into it): the reference's world stepped once with its own DiffContactHandler
    and once with the build's, its own LCP in both.  Contact pairs per step, final poses and d loss / d (box width, ball
    radius) agree."""
    from oracle import refshim
    refshim.install()
    from lcp_physics.physics.bodies import Circle, Rect
    from lcp_physics.physics.constraints import TotalConstraint
    from lcp_physics.physics.forces import Gravity
    from lcp_physics.physics.world import World

    def run(handler):
        wd = torch.tensor(80.0, dtype=torch.double, requires_grad=True)
        rad = torch.tensor(25.0, dtype=torch.double, requires_grad=True)
        floor = Rect([500, 600], [1000, 50], restitution=0.2, fric_coeff=0.6)
        box = Rect([0.3, 420, 520], torch.stack([wd, wd.new_tensor(50.0)]), restitution=0.2, fric_coeff=0.6)
        ball = Circle([560, 500], rad, vel=[0, -150, 0], restitution=0.2, fric_coeff=0.6)
        for b in (box, ball):
            b.add_force(Gravity(g=100))
        kw = {} if handler is None else {"contact_callback": handler}
        w = World([floor, box, ball], [TotalConstraint(floor)], dt=1.0 / 30, **kw)
        counts = []
        for _ in range(60):
            w.step()
            counts.append([(c[1], c[2]) for c in w.contacts])
        loss = (box.p ** 2).sum() + (ball.pos ** 2).sum()
        gw, gr = torch.autograd.grad(loss, [wd, rad])
        return counts, torch.cat([b.p for b in (floor, box, ball)]).detach().numpy(), float(gw), float(gr)

    c0, p0, gw0, gr0 = run(None)
    c1, p1, gw1, gr1 = run(emu_contact_handler())
    assert c0 == c1 and {(0, 1), (0, 2), (1, 2)} <= {pr for step in c0 for pr in step}     # box / floor, ball / floor, ball / box
    assert max(len(step) for step in c0) >= 3
    assert np.abs(p0 - p1).max() < 1e-9 * np.abs(p0).max()
    assert abs(gw0 - gw1) < 1e-7 * max(1.0, abs(gw0)) and abs(gr0 - gr1) < 1e-7 * max(1.0, abs(gr0)), (gw0, gw1, gr0, gr1)


def test_reference_3d_world_on_the_builds_lcp_kernels():
    """The same substitution under the reference's World3D (FWContactHandler, PdipmEngine): config 2's scene, one sphere
    dropped on the floor, 24 steps with time-of-contact differentials, against the rollout recorded with the
    reference's own solver (tests/golden/rollout_sphere.npz): trajectory and d sum|pos|^2 / d radius."""
    from oracle import refshim
    refshim.install()
    from oracle.gen import scenes
    from sdf_physics.physics3d.world import World3D

    g = np.load(os.path.join(GOLDEN, "rollout_sphere.npz"))
    bodies, joints, params = scenes.sphere_drop(seed=1, floor_dims=(4.0, 1.0, 4.0))
    calls = []
    w = World3D(bodies, joints, time_of_contact_diff=True)
    w.engine.lcp_solver = kernel_lcp_function(calls)
    for _ in range(24):
        w.step(fixed_dt=True)
    assert len(w.trajectory) == len(g["traj_t"]) and len(calls) > 0
    p = np.stack([b.p.detach().numpy() for b in bodies]); v = np.stack([b.v.detach().numpy() for b in bodies])
    assert np.abs(p - g["traj_p"][-1]).max() < 1e-8 and np.abs(v - g["traj_v"][-1]).max() < 1e-8
    loss = sum((b.pos ** 2).sum() for b in bodies)
    grad = torch.autograd.grad(loss, params)[0]
    assert abs(float(grad) - float(g["grad_0"])) < 1e-5 * abs(float(g["grad_0"])), (float(grad), float(g["grad_0"]))


def test_reference_3d_world_on_the_builds_contact_kernels():
    """Boundary B2' (contacts.py:21-26, looked up at world.py:52): the reference's World3D with its contact handler
    replaced by one that runs this build's broad + narrow phase (csrc/narrowphase.hip through the emulator) for the
    pair it is called with.  The kernel's contacts carry no autograd graph, so the trajectory is compared (a tilted box
    dropped on the floor: rejected attempts, dt halving, a time-of-contact event, sliding), not the gradient."""
    from oracle import refshim
    refshim.install()
    from oracle.gen import scenes
    from sdf_physics.physics3d.bodies import SDFBox, SDFCylinder
    from sdf_physics.physics3d.world import World3D
    from diffsdfsim_amd.engine import BatchEngine

    class KernelContactHandler:
        def __call__(self, args, geom1, geom2):
            if geom1 in geom2.no_contact:
                return
            world = args[0]
            ids = (geom1.body, geom2.body)
            bs = [world.bodies[i] for i in ids]
            one = lambda rows: np.stack([np.asarray(r, np.float64) for r in rows])[None]
            prm = [b.dims.detach().numpy() if isinstance(b, SDFBox) else
                   np.array([float(b.rad), float(b.height) if isinstance(b, SDFCylinder) else 0.0, 0.0]) for b in bs]
            spec = dict(pose=one([b.p.detach().numpy() for b in bs]), vel=np.zeros((1, 2, 6)), mass=np.ones((1, 2)),
                        inertia=np.tile(np.eye(3), (1, 2, 1, 1)), restitution=np.zeros((1, 2)), fric=np.zeros((1, 2)),
                        fext=np.zeros((1, 2, 6)), shape_prm=one(prm), mesh_id=np.arange(2, dtype=np.int32)[None],
                        shape_type=np.array([[0 if isinstance(b, SDFBox) else (2 if isinstance(b, SDFCylinder) else 1) for b in bs]], np.int32),
                        meshes=[(b.verts.detach().numpy(), b.faces.numpy()) for b in bs])
            E = BatchEngine(spec, backend=emu.EmuBackend(), eps=world.eps, tol=world.tol, fric_dirs=world.fric_dirs,
                            strict_no_pen=False)      # its constructor runs dss_find_contacts at the given poses
            t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.double)
            if int(E.get("invalid")[0]):                # penetration beyond tol: the world only needs to see that
                world.contacts.append(((t([0.0, 1.0, 0.0]), t(np.zeros(3)), t(np.zeros(3)), t(1.0)), ids[0], ids[1]))
                return
            n, body, geom = int(E.get("nc")[0]), E.get("c_body")[0], E.get("c_geom")[0]
            for c in range(n):
                world.contacts.append(((t(geom[0:3, c]), t(geom[3:6, c]), t(geom[6:9, c]), t(geom[9, c])),
                                       ids[body[0, c]], ids[body[1, c]]))

    g = np.load(os.path.join(GOLDEN, "rollout_boxdrop.npz"))
    bodies, joints, _params = scenes.box_drop(seed=7, requires_grad=False)
    w = World3D(bodies, joints, contact_callback=KernelContactHandler, time_of_contact_diff=True)
    for _ in range(12):
        w.step(fixed_dt=True)
    assert len(w.trajectory) == len(g["traj_t"])
    p = np.stack([b.p.detach().numpy() for b in bodies]); v = np.stack([b.v.detach().numpy() for b in bodies])
    assert np.abs(p - g["traj_p"][-1]).max() < 1e-7 and np.abs(v - g["traj_v"][-1]).max() < 1e-7


def test_reference_3d_world_on_the_builds_engine_plugin():
    """Boundary B2 (engines.py:16-19, looked up at world.py:51): the reference's World3D with `engine=` an Engine whose
    solve_dynamics(world, dt) hands the world's state and contact list to dss_solve_dynamics (assembly + the
    contact-structured LCP, csrc/step.hip + lcp_contact.hip through the emulator) and returns the new velocities.
    Values only, so the trajectory is compared: a box on the floor shoved sideways until friction saturates."""
    import ctypes
    from oracle import refshim
    refshim.install()
    from oracle.gen import scenes
    from sdf_physics.physics3d.world import World3D
    from diffsdfsim_amd.engine import BatchEngine

    class KernelEngine:
        def __init__(self):
            self.E = None

        def solve_dynamics(self, world, dt):
            bs = world.bodies
            nb = len(bs)
            one = lambda rows: np.stack([np.asarray(r, np.float64) for r in rows])[None]
            if self.E is None:
                Je = world.Je().detach().numpy()[None]
                spec = dict(pose=one([b.p.detach().numpy() for b in bs]), vel=np.zeros((1, nb, 6)), mass=np.ones((1, nb)),
                            inertia=np.tile(np.eye(3), (1, nb, 1, 1)), restitution=np.zeros((1, nb)), fric=np.zeros((1, nb)),
                            fext=np.zeros((1, nb, 6)), shape_prm=one([b.dims.detach().numpy() for b in bs]),
                            shape_type=np.zeros((1, nb), np.int32), mesh_id=np.arange(nb, dtype=np.int32)[None],
                            meshes=[(b.verts.detach().numpy(), b.faces.numpy()) for b in bs], Je=Je)
                self.E = BatchEngine(spec, backend=emu.EmuBackend(), eps=world.eps, tol=world.tol, fric_dirs=world.fric_dirs,
                                     strict_no_pen=False, maxc=64)
            E = self.E
            A = E.arr
            A["pose"][...] = one([b.p.detach().numpy() for b in bs]); A["vel"][...] = one([b.v.detach().numpy() for b in bs])
            A["mass"][...] = [[float(b.mass) for b in bs]]
            A["inertia"][...] = one([b.ang_inertia.detach().numpy().reshape(9) for b in bs])
            A["restitution"][...] = [[float(b.restitution) for b in bs]]; A["fric"][...] = [[float(b.fric_coeff) for b in bs]]
            A["fext"][...] = world.apply_forces(world.t).detach().numpy().reshape(1, nb, 6)
            n = len(world.contacts)
            A["nc"][0] = n
            for c, ((nr, p1, p2, pen), i1, i2) in enumerate(world.contacts):
                A["c_body"][0, :, c] = (i1, i2)
                A["c_geom"][0, 0:3, c] = nr.detach().numpy(); A["c_geom"][0, 3:6, c] = p1.detach().numpy()
                A["c_geom"][0, 6:9, c] = p2.detach().numpy(); A["c_geom"][0, 9, c] = float(pen)
            A["dt_try"][0] = float(dt); A["active"][0] = 1
            rc = E.be.lib.dss_solve_dynamics(ctypes.byref(E.W), ctypes.c_void_p(E.be.ptr(E.lcp_ws)),
                                             ctypes.c_size_t(E.lcp_ws_bytes), E.be.stream())
            A["active"][0] = 0
            assert rc == 0
            return torch.tensor(-A["x"][0].copy())

    g = np.load(os.path.join(GOLDEN, "rollout_stack1.npz"))
    bodies, joints, _params = scenes.box_stack(nbox=1, seed=3, vel_scale=1.0, push=2.0, requires_grad=False)
    w = World3D(bodies, joints, engine=KernelEngine, time_of_contact_diff=True)
    for _ in range(4):
        w.step(fixed_dt=True)
    assert len(w.trajectory) == len(g["traj_t"])
    p = np.stack([b.p.detach().numpy() for b in bodies]); v = np.stack([b.v.detach().numpy() for b in bodies])
    assert np.abs(p - g["traj_p"][-1]).max() < 1e-8 and np.abs(v - g["traj_v"][-1]).max() < 1e-8
