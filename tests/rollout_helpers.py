"""Build a BatchEngine spec from a rollout golden and compare trajectories (shared by emu + gpu tests)."""
import os

import numpy as np

from helpers import GOLDEN


def load_rollout(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def spec_from_golden(g, copies=1, level_set_mesh=None):
    """Scene replicated `copies` times along the batch axis (identical scenes must give identical results).
    Bodies whose (level-set) mesh is not in the golden get it from `level_set_mesh(body index)`."""
    nb = len(g["mass"])
    meshes = [(g["verts_%d" % i], g["faces_%d" % i]) if "verts_%d" % i in g else level_set_mesh(i) for i in range(nb)]
    rep = lambda a: np.repeat(np.asarray(a)[None], copies, axis=0)
    Je = np.zeros((6 * len(g["fixed"]), 6 * nb))
    for k, b in enumerate(g["fixed"]):
        Je[6 * k:6 * k + 6, 6 * b:6 * b + 6] = np.eye(6)   # TotalConstraint3D.J() = I_6 (constraints.py:131-137)
    extra = dict(shape_aux=rep(g["shape_aux"])) if "shape_aux" in g else {}
    if "no_contact" in g:
        extra["no_contact"] = np.asarray(g["no_contact"], np.uint8)
    return dict(extra, pose=rep(g["pose0"]), vel=rep(g["vel0"]), mass=rep(g["mass"]), inertia=rep(g["inertia"]),
                restitution=rep(g["restitution"]), fric=rep(g["fric"]), fext=rep(g["fext"]),
                shape_type=rep(g["shape_type"]), shape_prm=rep(g["shape_prm"]), mesh_id=rep(np.arange(nb)),
                meshes=meshes, Je=rep(Je))


def pair_multiset(body, n):
    return sorted((int(a), int(b)) for a, b in body[:n])


def engine_kwargs(g, **over):
    kw = dict(dt=float(g["dt"]), eps=float(g["eps"]), tol=float(g["tol"]), fric_dirs=int(g["fric_dirs"]),
              toc_diff=bool(g["toc_diff"]), maxc=64, max_cand=1024, max_pc=32)
    if "strict_no_pen" in g:
        kw["strict_no_pen"] = bool(g["strict_no_pen"])
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


def param_grads(E, g, s=0):
    """Chain the engine-level gradients onto the reference's leaf parameters (dims / radius)."""
    adj = {k: E.be.to_numpy(v) for k, v in E.adj.items()}
    out = []
    nb = len(g["mass"])
    pi = 0
    for b in range(nb):
        if b in g["fixed"]:
            continue
        gI = adj["g_inertia"][s, b].reshape(3, 3)
        m = g["mass"][b]
        if g["shape_type"][b] == 0:
            d = g["shape_prm"][b]
            # custom box inertia m/12 diag(d1^2+d2^2, d0^2+d2^2, d0^2+d1^2) (bodies.py:796-797)
            dI = np.array([(gI[1, 1] + gI[2, 2]) * 2 * d[0], (gI[0, 0] + gI[2, 2]) * 2 * d[1], (gI[0, 0] + gI[1, 1]) * 2 * d[2]]) * m / 12
            out.append(adj["g_prm"][s, b] + dI)
        elif g["shape_type"][b] == 2:
            r, h = g["shape_prm"][b, 0], g["shape_prm"][b, 1]
            # custom cylinder inertia m diag(a, a, r^2/2), a = (3 r^2 + h^2)/12 (bodies.py:925-927)
            out.append(np.array(adj["g_prm"][s, b, 0] + m * (0.5 * r * (gI[0, 0] + gI[1, 1]) + r * gI[2, 2])))
            out.append(np.array(adj["g_prm"][s, b, 1] + m * h / 6 * (gI[0, 0] + gI[1, 1])))
        else:
            r = g["shape_prm"][b, 0]
            out.append(np.array(adj["g_prm"][s, b, 0] + 0.8 * m * r * np.trace(gI)))  # 2/5 m r^2 I (bodies.py:993-994)
        pi += 1
    return out


def check_gradients(E, g, tol=1e-5, s=0):
    """Reference gradients are bimodal on flat-on-flat contacts (see oracle/gen/gen_rollout_golden.py:
    `stable_mask` compares two rounding-noise Laplacians); both branches are in the golden and the kernel
    must reproduce one of them."""
    got = param_grads(E, g, s)
    errs = []
    for key in ("grad_%d", "gradB_%d"):
        e = 0.0
        for i, gi in enumerate(got):
            want = g[key % i]
            e = max(e, np.abs(gi - want).max() / max(np.abs(want).max(), 1e-300))
        errs.append(e)
    assert min(errs) < tol, (errs, got)


def rollout_and_sweep(E, nsteps):
    for _ in range(nsteps):
        E.step()
    adj = E._adjoint()
    ap = np.zeros_like(E.get("pose"))
    ap[:, :, 4:] = 2 * E.get("pose")[:, :, 4:]     # d/dpos of sum |pos|^2
    adj["a_pose"][...] = E.be.from_numpy(ap)
    adj["cur_slot"][...] = E.be.from_numpy((E.get("nsub") - 1).astype(np.int32))
    adj["lo_slot"][...] = -1      # down to and including the contacts found at construction
    E.backward_sweep(int(E.get("nsub").max()) + 1)


