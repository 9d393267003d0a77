"""The reference's 2-D world (BASELINE configs[0]: `lcp_physics.physics` -- Circle / Rect / Hull bodies, TotalConstraint, Gravity,
`World.step`) on the device library: contacts of the body pairs of a step from `dss_contacts2d_forward` (one launch for all pairs
unless a polygon appears in several, see `find_contacts`), the mixed
LCP of `PdipmEngine.solve_dynamics` from `dss_lcp_dense_forward` / `_backward` (diffsdfsim_amd.lcp.LCPFunction), state and
Jacobian assembly as torch tensors on the device, gradients by autograd through those three.

What is restated, with the reference's semantics:
  bodies      p = [theta, x, y], v = [omega, vx, vy], M = diag(I, m, m); Circle I = m r^2 / 2; Hull (vertices about the
              centroid, polygon inertia); Rect I = m (w^2 + h^2) / 12                      lcp_physics/physics/bodies.py:32-323
  forces      Gravity = [0, 0, 1] m g (y points down)                                      forces.py:55-72
  constraints TotalConstraint: the three identity rows of Je                               constraints.py:196-211
  step        step / step_dt: solve, move, detect, accept or halve dt (strict_no_penetration) world.py:119-139, 241-379
  Jacobians   Jc, Jf (two directions), mu, E, restitutions                                world.py:402-501
  engine      u = M v + dt f; contact-free: the KKT inverse; else the LCP                  engines.py:31-83
Not restated: joints other than TotalConstraint, the time-of-contact differential (off by default in 2-D), post-stabilisation,
pygame drawing.  A world holds its pairs in the order (i < j), bodies in list order -- the order the goldens were recorded in
(oracle/refshim/fake_ode.py)."""
import torch

from ..lcp import LCPFunction
from .contacts import MAXV, contacts2d


class Defaults:
    DIM, EPSILON, TOL, RESTITUTION, FRIC_COEFF, FRIC_DIRS, FPS = 2, 0.1, 1e-6, 0.5, 0.9, 2, 30
    DT = 1.0 / FPS
    DTYPE = torch.double
    DEVICE = torch.device("cuda:0")


def _t(x):
    if torch.is_tensor(x):
        return x.to(device=Defaults.DEVICE, dtype=Defaults.DTYPE)
    return torch.tensor(x, dtype=Defaults.DTYPE, device=Defaults.DEVICE)


class Gravity:
    def __init__(self, g=10.0):
        self.g = g

    def force(self, body, t):
        return torch.stack([body.mass.new_zeros(()), body.mass.new_zeros(()), body.mass * self.g])


class Body:
    kind = -1

    def __init__(self, pos, vel=(0, 0, 0), mass=1, restitution=Defaults.RESTITUTION, fric_coeff=Defaults.FRIC_COEFF, eps=Defaults.EPSILON):
        pos, vel = _t(pos), _t(vel)
        self.p = torch.cat([pos.new_zeros(1), pos]) if pos.numel() == 2 else pos
        self.v = torch.cat([vel.new_zeros(1), vel]) if vel.numel() == 2 else vel
        self.mass, self.restitution, self.fric_coeff = _t(mass), _t(restitution), _t(fric_coeff)
        self.forces, self.no_contact = [], set()
        self.ang_inertia = self._ang_inertia()

    pos = property(lambda self: self.p[1:])
    rot = property(lambda self: self.p[0:1])

    def M(self):
        return torch.diag(torch.stack([self.ang_inertia.reshape(()), self.mass, self.mass]))

    def add_force(self, f):
        self.forces.append(f)

    def add_no_contact(self, other):
        self.no_contact.add(other)
        other.no_contact.add(self)

    def apply_forces(self, t):
        if not self.forces:
            return self.v.new_zeros(3)
        return sum(f.force(self, t) for f in self.forces)


class Circle(Body):
    kind = 0

    def __init__(self, pos, rad, **kw):
        self.rad = _t(rad)
        super().__init__(pos, **kw)

    def _ang_inertia(self):
        return self.mass * self.rad * self.rad / 2


class Hull(Body):
    kind = 1

    def __init__(self, ref_point, vertices, **kw):
        v = torch.stack([_t(x) for x in vertices])
        nxt = torch.roll(v, -1, 0)
        if len(v) < 3 or len(v) > MAXV or float(((nxt[:, 0] - v[:, 0]) * (nxt[:, 1] + v[:, 1])).sum().detach()) >= 0:
            raise ValueError("a hull needs 3..%d vertices in clockwise order (y pointing down)" % MAXV)
        cr = nxt[:, 0] * v[:, 1] - nxt[:, 1] * v[:, 0]
        centroid = (cr[:, None] * (v + nxt)).sum(0) / (6.0 * (cr / 2).sum())
        self.verts0 = v - centroid           # about the centroid, at the angle the body is created with
        self.last_sat_idx = 0
        super().__init__(_t(ref_point)[-2:] + centroid, **kw)
        self.rot0 = self.p[0].detach().clone()

    def verts(self):
        a = self.p[0] - self.rot0
        c, s = torch.cos(a), torch.sin(a)
        R = torch.stack([torch.stack([c, -s]), torch.stack([s, c])])
        return self.verts0 @ R.t()

    def _ang_inertia(self):
        v, nxt = self.verts0, torch.roll(self.verts0, -1, 0)
        nc = (nxt[:, 0] * v[:, 1] - nxt[:, 1] * v[:, 0]).abs()
        num = (nc * ((v * v).sum(1) + (v * nxt).sum(1) + (nxt * nxt).sum(1))).sum()
        return self.mass * num / (6.0 * nc.sum())


class Rect(Hull):
    def __init__(self, pos, dims, **kw):
        self.dims = _t(dims)
        pos = _t(pos)
        h = self.dims / 2
        sg = h.new_tensor([[1.0, 1.0], [-1.0, 1.0], [-1.0, -1.0], [1.0, -1.0]])
        super().__init__(pos[-2:], list(sg * h), **kw)
        if pos.numel() == 3:                 # created turned: the vertices are those of the unturned rectangle at angle 0
            self.p = torch.cat([pos[0:1], self.p[1:]])
            self.rot0 = self.p.new_zeros(())

    def _ang_inertia(self):
        return self.mass * (self.dims ** 2).sum() / 12


class TotalConstraint:
    num_constraints = 3

    def __init__(self, body1):
        self.body1 = body1


class World:
    """`lcp_physics.physics.world.World(bodies, constraints, dt, eps, tol, fric_dirs, strict_no_penetration)` for the bodies
    above.  `contacts` after a step: list of ((normal, p1, p2, penetration), i1, i2), as the reference keeps it."""

    def __init__(self, bodies, constraints=(), dt=Defaults.DT, eps=Defaults.EPSILON, tol=Defaults.TOL, fric_dirs=Defaults.FRIC_DIRS,
                 strict_no_penetration=True, max_iter=10):
        if fric_dirs != 2:
            raise NotImplementedError("the 2-D world has two friction directions (world.py:446-472)")
        self.bodies, self.dt, self.eps, self.tol, self.fric_dirs = list(bodies), dt, eps, tol, fric_dirs
        self.strict_no_pen, self.max_iter = strict_no_penetration, max_iter
        self.t, self.trajectory, self.lcp_calls = 0.0, [], 0
        nb = len(self.bodies)
        for c in constraints:
            if not isinstance(c, TotalConstraint):
                raise NotImplementedError("only TotalConstraint is built (SURVEY.md section 2: joints are out of scope)")
        self.pinned = [self.bodies.index(c.body1) for c in constraints]
        self._M = torch.block_diag(*[b.M() for b in self.bodies])
        self._Je = self._M.new_zeros(3 * len(self.pinned), 3 * nb)
        for r, i in enumerate(self.pinned):
            self._Je[3 * r:3 * r + 3, 3 * i:3 * i + 3] = torch.eye(3, dtype=self._M.dtype, device=self._M.device)
        self.pairs = [(i, j) for i in range(nb) for j in range(i + 1, nb) if self.bodies[j] not in self.bodies[i].no_contact]
        self.v = torch.cat([b.v for b in self.bodies])
        self._kkt_inv = None
        self.contacts = []
        self.find_contacts()
        if self.strict_no_pen and any(float(c[0][3].detach()) > self.tol for c in self.contacts):
            raise AssertionError("Interpenetration at start")

    # -- state ---------------------------------------------------------------------------------------------------------------
    def set_v(self, v):
        self.v = v
        for i, b in enumerate(self.bodies):
            b.v = v[3 * i:3 * i + 3]

    def set_p(self, p):
        for i, b in enumerate(self.bodies):
            b.p = p[3 * i:3 * i + 3]

    def get_p(self):
        return torch.cat([b.p for b in self.bodies])

    # -- contacts ----------------------------------------------------------------------------------------------------------
    def find_contacts(self):
        """Contacts of all pairs, in pair order.  Pairs go to the kernel in one launch as long as no polygon appears twice: a
        polygon's `last_sat_idx` is state that the reference carries from one pair to the next (contacts.py:124-131, 153-158),
        so a pair that meets a polygon again starts a new launch with the index the earlier pair left."""
        self.contacts = []
        group, seen = [], set()
        for pr in self.pairs:
            hulls = {i for i in pr if self.bodies[i].kind == 1}
            if hulls & seen:
                self._detect(group)
                group, seen = [], set()
            group.append(pr)
            seen |= hulls
        if group:
            self._detect(group)

    def _detect(self, pairs):
        dev = self._M.device
        P = len(pairs)
        zero = self._M.new_zeros(())
        padv = self._M.new_zeros(MAXV, 2)
        rows = {k: ([], []) for k in ("pos", "rad", "verts")}
        meta = torch.zeros(3, 2, P, dtype=torch.int32)        # kind, nv, sat
        vcache = {}
        for q, pr in enumerate(pairs):
            for s, i in enumerate(pr):
                b = self.bodies[i]
                rows["pos"][s].append(b.pos)
                rows["rad"][s].append(b.rad.reshape(()) if b.kind == 0 else zero)
                if b.kind == 1:
                    if i not in vcache:
                        v = b.verts()
                        vcache[i] = torch.cat([v, padv[:MAXV - len(v)]])
                    rows["verts"][s].append(vcache[i])
                    meta[1, s, q], meta[2, s, q] = len(b.verts0), b.last_sat_idx
                else:
                    rows["verts"][s].append(padv)
                meta[0, s, q] = b.kind
        st = lambda k: torch.stack([torch.stack(rows[k][0]), torch.stack(rows[k][1])])      # noqa: E731
        meta = meta.to(dev)
        out, count, sat = contacts2d(st("pos"), st("rad"), st("verts"), meta[0].contiguous(), meta[1].contiguous(),
                                     meta[2].contiguous(), self.eps)
        count, sat = count.cpu(), sat.cpu()
        for q, (i, j) in enumerate(pairs):
            for s, k in enumerate((i, j)):
                if self.bodies[k].kind == 1:
                    self.bodies[k].last_sat_idx = int(sat[s, q])
            for c in range(int(count[q])):
                o = out[q, c]
                self.contacts.append(((o[0:2], o[2:4], o[4:6], o[6]), i, j))

    # -- Jacobians, vectorised over the contacts -----------------------------------------------------------------------------------
    def _contact_rows(self):
        nc, nb = len(self.contacts), len(self.bodies)
        n = torch.stack([c[0][0] for c in self.contacts])
        p1 = torch.stack([c[0][1] for c in self.contacts])
        p2 = torch.stack([c[0][2] for c in self.contacts])
        i1 = torch.tensor([c[1] for c in self.contacts], device=n.device)
        i2 = torch.tensor([c[2] for c in self.contacts], device=n.device)
        cross = lambda a, b: a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]      # noqa: E731

        def rows(d):             # [nc, 3 nb]: +[p1 x d, d] on body 1, -[p2 x d, d] on body 2
            J = n.new_zeros(nc, nb, 3)
            ar = torch.arange(nc, device=n.device)
            J[ar, i1] = torch.cat([cross(p1, d)[:, None], d], 1)
            J[ar, i2] = -torch.cat([cross(p2, d)[:, None], d], 1)
            return J.reshape(nc, 3 * nb)
        d1 = torch.stack([n[:, 1], -n[:, 0]], 1)
        Jc = rows(n)
        Jf = torch.stack([rows(d1), rows(-d1)], 1).reshape(2 * nc, 3 * nb)
        fr = torch.stack([b.fric_coeff for b in self.bodies])
        re = torch.stack([b.restitution for b in self.bodies])
        mu = torch.diag(0.5 * (fr[i1] + fr[i2]))
        E = torch.kron(torch.eye(nc, dtype=n.dtype, device=n.device), n.new_ones(2, 1))
        return Jc, Jf, mu, E, 0.5 * (re[i1] + re[i2])

    # -- PdipmEngine.solve_dynamics (engines.py:31-83) --------------------------------------------------------------------------------
    def solve_dynamics(self, dt):
        M, Je, nz = self._M, self._Je, self._M.shape[0]
        neq = Je.shape[0]
        f = torch.cat([b.apply_forces(self.t) for b in self.bodies])
        u = M @ self.v + dt * f
        if not self.contacts:
            if self._kkt_inv is None:
                P = torch.cat([torch.cat([M, -Je.t()], 1), torch.cat([Je, Je.new_zeros(neq, neq)], 1)]) if neq else M
                self._kkt_inv = torch.inverse(P)
            return (self._kkt_inv @ torch.cat([u, u.new_zeros(neq)]))[:nz]
        Jc, Jf, mu, E, rest = self._contact_rows()
        nc = Jc.shape[0]
        G = torch.cat([Jc, Jf, Jf.new_zeros(nc, nz)])
        F = G.new_zeros(G.shape[0], G.shape[0])
        F[nc:3 * nc, 3 * nc:] = E
        F[3 * nc:, :nc] = mu
        F[3 * nc:, nc:3 * nc] = -E.t()
        h = torch.cat([(Jc @ self.v) * rest, Jc.new_zeros(3 * nc)])
        if neq:
            A, b = Je[None], Je.new_zeros(1, neq)
        else:
            A, b = Je.new_zeros(0), Je.new_zeros(0)
        self.lcp_calls += 1
        x = -LCPFunction(max_iter=self.max_iter, verbose=-1)(M[None], u[None], G[None], h[None], A, b, F[None])
        return x[0, :nz]

    # -- World.step / step_dt (world.py:119-139, 241-379) ----------------------------------------------------------------------------
    def step(self, fixed_dt=False):
        had = False
        if fixed_dt:
            end_t = self.t + self.dt
            while self.t < end_t:
                self.step_dt(end_t - self.t)
                had = had or bool(self.contacts)
        else:
            self.step_dt(self.dt)
            had = bool(self.contacts)
        return had

    def step_dt(self, dt):
        start_p, start_v, start_contacts = self.get_p(), self.v, self.contacts
        while True:
            self.set_v(self.solve_dynamics(dt))
            self.set_p(self.get_p() + self.v * dt)
            self.find_contacts()
            if all(float(c[0][3].detach()) <= self.tol for c in self.contacts):
                break
            if not self.strict_no_pen and dt < self.dt / 2 ** 10:
                break
            dt = dt / 2
            self.set_p(start_p)
            self.set_v(start_v)
            self.contacts = start_contacts
        self.trajectory.append((self.t, self.get_p(), self.v, self.contacts))
        self.t += dt


def run_world(world, run_time=10.0):
    """`run_world(world, run_time=...)` without a screen (world.py:513-587)."""
    while world.t < run_time:
        world.step()
