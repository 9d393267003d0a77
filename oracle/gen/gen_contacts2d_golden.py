"""Golden vectors of the reference's 2-D analytic contact handler (DiffContactHandler, lcp_physics/physics/contacts.py:55-357)
for random pairs of circles and convex polygons: the contact tuples it appends, the `last_sat_idx` it leaves on the bodies and,
from autograd, the gradient of a random linear functional of the tuples w.r.t. positions, radii and vertices
-> tests/golden/contacts2d.npz.  Run in the build container only:  python -m oracle.gen.gen_contacts2d_golden
"""
import math
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refshim  # noqa: E402

refshim.install()
from lcp_physics.physics.bodies import Circle, Hull  # noqa: E402
from lcp_physics.physics.contacts import DiffContactHandler  # noqa: E402

MAXV = 8
EPS = 0.1


def polygon(rng, nv, size):
    """Convex polygon in the orientation the reference asserts (bodies.py:222-229), vertices on an ellipse."""
    ang = np.sort(rng.uniform(0, 2 * math.pi, nv))
    while np.min(np.diff(np.concatenate([ang, [ang[0] + 2 * math.pi]]))) < 0.35:
        ang = np.sort(rng.uniform(0, 2 * math.pi, nv))
    a, b, rot = size * rng.uniform(0.6, 1.0), size * rng.uniform(0.6, 1.0), rng.uniform(0, math.pi)
    v = np.stack([a * np.cos(ang), b * np.sin(ang)], 1)
    v = v @ np.array([[math.cos(rot), -math.sin(rot)], [math.sin(rot), math.cos(rot)]]).T
    tot = sum((v[(i + 1) % nv][0] - v[i][0]) * (v[(i + 1) % nv][1] + v[i][1]) for i in range(nv))
    return v if tot < 0 else v[::-1].copy()


class World:
    def __init__(self, bodies):
        self.bodies, self.eps, self.contacts = bodies, EPS, []


def make_body(rng, kind, pos):
    if kind == 0:
        b = Circle(list(pos), float(rng.uniform(0.3, 1.0)))
    else:
        nv = int(rng.integers(3, 7))
        b = Hull(list(pos), [list(p) for p in polygon(rng, nv, rng.uniform(0.5, 1.2))])
        b.last_sat_idx = int(rng.integers(0, nv))
    return b


def extent(b, d):
    if isinstance(b, Circle):
        return float(b.rad)
    return max(float(v[0] * d[0] + v[1] * d[1]) for v in b.verts)


def case(rng, kinds):
    b1 = make_body(rng, kinds[0], rng.uniform(-1, 1, 2))
    b2 = make_body(rng, kinds[1], [0.0, 0.0])
    th = rng.uniform(0, 2 * math.pi)
    d = np.array([math.cos(th), math.sin(th)])
    mode = rng.uniform()
    gap = rng.uniform(-0.25, 0.3) if mode < 0.8 else rng.uniform(-1.5, -0.3)      # mostly near touching, some deep
    dist = max(extent(b1, -d) + extent(b2, d) + gap, 0.02)
    p2 = np.array([float(b1.pos[0]), float(b1.pos[1])]) - 0 * d
    # body 1 sits at distance `dist` from body 2 along d
    leaf = lambda x: torch.tensor(np.asarray(x, dtype=np.float64), requires_grad=True)   # noqa: E731
    pos2 = leaf(rng.uniform(-1, 1, 2))
    pos1 = leaf(pos2.detach().numpy() + dist * d)
    bodies = [b1, b2]
    leaves = []
    for b, pos in zip(bodies, (pos1, pos2)):
        b.pos = pos
        L = {"pos": pos, "rad": None, "verts": None}
        if isinstance(b, Circle):
            b.rad = leaf(float(b.rad))
            L["rad"] = b.rad
        else:
            b.verts = [leaf(v.detach().numpy()) for v in b.verts]
            L["verts"] = b.verts
        leaves.append(L)
    sat_in = [0 if isinstance(b, Circle) else b.last_sat_idx for b in bodies]
    w = World(bodies)

    class G:
        pass
    g1, g2 = G(), G()
    g1.body, g2.body, g1.no_contact, g2.no_contact = 0, 1, set(), set()
    DiffContactHandler()([w], g1, g2)
    out = np.zeros((2, 7)); gout = rng.normal(size=(2, 7))
    loss = 0.0
    for q, (c, i1, i2) in enumerate(w.contacts):
        assert (i1, i2) == (0, 1)
        flat = torch.cat([c[0].reshape(-1), c[1].reshape(-1), c[2].reshape(-1), c[3].reshape(-1)])
        out[q] = flat.detach().numpy()
        loss = loss + (flat * torch.tensor(gout[q])).sum()
    g_pos, g_rad, g_verts = np.zeros((2, 2)), np.zeros(2), np.zeros((2, MAXV, 2))
    if len(w.contacts):
        loss.backward()
        for s, L in enumerate(leaves):
            if L["pos"].grad is not None:
                g_pos[s] = L["pos"].grad.numpy()
            if L["rad"] is not None and L["rad"].grad is not None:
                g_rad[s] = float(L["rad"].grad)
            if L["verts"] is not None:
                for i, v in enumerate(L["verts"]):
                    if v.grad is not None:
                        g_verts[s, i] = v.grad.numpy()
    verts = np.zeros((2, MAXV, 2)); nv = np.zeros(2, np.int32)
    for s, b in enumerate(bodies):
        if not isinstance(b, Circle):
            nv[s] = len(b.verts)
            verts[s, :nv[s]] = np.stack([v.detach().numpy() for v in b.verts])
    return dict(kind=np.array(kinds, np.int32), nv=nv, pos=np.stack([pos1.detach().numpy(), pos2.detach().numpy()]),
                rad=np.array([float(b.rad) if isinstance(b, Circle) else 0.0 for b in bodies]), verts=verts,
                sat_in=np.array(sat_in, np.int32),
                sat_out=np.array([0 if isinstance(b, Circle) else b.last_sat_idx for b in bodies], np.int32),
                count=np.int32(len(w.contacts)), out=out, gout=gout, g_pos=g_pos, g_rad=g_rad, g_verts=g_verts)


def main():
    rng = np.random.default_rng(7)
    random.seed(7)
    torch.set_default_dtype(torch.float64)
    cases = []
    for kinds, n in (((0, 0), 40), ((0, 1), 90), ((1, 0), 90), ((1, 1), 180)):
        for _ in range(n):
            cases.append(case(rng, kinds))
    d = {k: np.stack([c[k] for c in cases]) for k in cases[0]}
    d["eps"] = np.float64(EPS)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "contacts2d.npz"), **d)
    cnt = d["count"]
    print("%d pairs: %d without contact, %d with one, %d with two" % (len(cnt), (cnt == 0).sum(), (cnt == 1).sum(), (cnt == 2).sum()))
    for kinds in ((0, 0), (0, 1), (1, 0), (1, 1)):
        m = (d["kind"] == np.array(kinds)).all(1)
        print(kinds, "counts", np.bincount(cnt[m], minlength=3))


if __name__ == "__main__":
    main()
