"""Build a BatchEngine spec from a rollout golden and compare trajectories (shared by emu + gpu tests)."""
import os

import numpy as np

from helpers import GOLDEN


def load_rollout(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def spec_from_golden(g, copies=1):
    """Scene replicated `copies` times along the batch axis (identical scenes must give identical results)."""
    nb = len(g["mass"])
    meshes = [(g["verts_%d" % i], g["faces_%d" % i]) for i in range(nb)]
    rep = lambda a: np.repeat(np.asarray(a)[None], copies, axis=0)
    Je = np.zeros((6 * len(g["fixed"]), 6 * nb))
    for k, b in enumerate(g["fixed"]):
        Je[6 * k:6 * k + 6, 6 * b:6 * b + 6] = np.eye(6)   # TotalConstraint3D.J() = I_6 (constraints.py:131-137)
    return dict(pose=rep(g["pose0"]), vel=rep(g["vel0"]), mass=rep(g["mass"]), inertia=rep(g["inertia"]),
                restitution=rep(g["restitution"]), fric=rep(g["fric"]), fext=rep(g["fext"]),
                shape_type=rep(g["shape_type"]), shape_prm=rep(g["shape_prm"]), mesh_id=rep(np.arange(nb)),
                meshes=meshes, Je=rep(Je))


def pair_multiset(body, n):
    return sorted((int(a), int(b)) for a, b in body[:n])


def engine_kwargs(g, **over):
    kw = dict(dt=float(g["dt"]), eps=float(g["eps"]), tol=float(g["tol"]), fric_dirs=int(g["fric_dirs"]),
              toc_diff=bool(g["toc_diff"]), maxc=64, max_cand=1024, max_pc=32)
    kw.update(over)
    return kw


def check_contacts(E, s, body_ref, geom_ref, n_ref, tol=1e-6):
    """Same ordered pair list; per ordered pair the same set of contact points (order inside a pair is
    implementation defined on both sides: Qhull vertex order vs ascending face id)."""
    nc = int(E.get("nc")[s])
    body = E.get("c_body")[s][:, :nc].T
    geom = E.get("c_geom")[s][:, :nc].T
    assert nc == n_ref, (nc, n_ref)
    assert [tuple(r) for r in body] == [tuple(r) for r in body_ref[:n_ref]], "ordered pair list differs"
    for pair in sorted(set(map(tuple, body))):
        m = (body == pair).all(axis=1); mr = (body_ref[:n_ref] == pair).all(axis=1)
        a = geom[m]; b = geom_ref[:n_ref][mr]
        ia = np.lexsort(np.round(a[:, 3:6], 6).T[::-1]); ib = np.lexsort(np.round(b[:, 3:6], 6).T[::-1])
        assert np.abs(a[ia] - b[ib]).max() < tol, (pair, np.abs(a[ia] - b[ib]).max())
