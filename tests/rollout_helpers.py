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
    gb = [i for i in range(nb) if "grid_%d" % i in g]      # voxel-grid SDF bodies
    if gb:
        extra["grids"] = [g["grid_%d" % i] for i in gb]
        extra["grid_id"] = rep(np.array([gb.index(i) if i in gb else -1 for i in range(nb)], np.int32))
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
    if "grad_flags" in g:
        kw["grad_flags"] = int(g["grad_flags"])
    kw.update(over)
    return kw


def check_contacts(E, s, body_ref, geom_ref, n_ref, tol=1e-6, coin_tol=None, corner_ties=False):
    """Same ordered pair list; per ordered pair the same set of contact points (order inside a pair is
    implementation defined on both sides: Qhull vertex order vs ascending face id).  `coin_tol`: looser bound for the NORMAL
    (columns 0..2) -- a flat-on-flat contact takes it from either body on a rounding-noise comparison (contacts.py:198) and the
    two bodies' normals may be up to the cluster angle (1e-2 rad) apart.  `corner_ties`: a pair of boxes lying flat on each other may
    differ from the reference by ONE contact (the corner of the face or its two neighbours on the edges become hull vertices,
    decided by an argmin over three equal vertex distances, contacts.py:57-61: the reference's own count for such a pair flickers
    from step to step, test_bench_scenes_gpu._check_tape_against_golden); its points are then not compared.  Returns the number
    of such pairs."""
    nc = int(E.get("nc")[s])
    body = E.get("c_body")[s][:, :nc].T
    geom = E.get("c_geom")[s][:, :nc].T
    ties = 0
    if corner_ties:
        dd = lambda L: [p for i, p in enumerate(L) if i == 0 or p != L[i - 1]]
        assert dd([tuple(r) for r in body]) == dd([tuple(r) for r in body_ref[:n_ref]]), "ordered pair list differs"
    else:
        assert nc == n_ref, (nc, n_ref)
        assert [tuple(r) for r in body] == [tuple(r) for r in body_ref[:n_ref]], "ordered pair list differs"
    for pair in sorted(set(map(tuple, body))):
        m = (body == pair).all(axis=1); mr = (body_ref[:n_ref] == pair).all(axis=1)
        a = geom[m]; b = geom_ref[:n_ref][mr]
        if corner_ties and len(a) != len(b):
            assert abs(len(a) - len(b)) == 1, (pair, len(a), len(b))
            ties += 1
            continue
        ia = np.lexsort(np.round(a[:, 3:6], 6).T[::-1]); ib = np.lexsort(np.round(b[:, 3:6], 6).T[::-1])
        d = np.abs(a[ia] - b[ib])
        assert d[:, 3:].max() < tol and d[:, :3].max() < (coin_tol or tol), (pair, d.max())
    return ties


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


def check_gradients(E, g, tol=1e-5, s=0, branch_tol=1e-5):
    """Reference gradients are bimodal on flat-on-flat contacts (see oracle/gen/gen_rollout_golden.py:
    `stable_mask` compares two rounding-noise Laplacians); both branches are in the golden.  Where the golden records the
    reference's choice per contact, the build's choices are compared with it (check_branches_and_pick_reference): the same
    branch is REQUIRED wherever the margin exceeds noise, and if the build took exactly the choices of one of the two
    recorded runs its gradient must equal that run's to `branch_tol` (north_star: 1e-5).  Otherwise (choices mixed between
    the two runs on coin-flip contacts) it must be within `tol` of the nearer one."""
    got = param_grads(E, g, s)
    errs = {}
    for tag, key in (("A", "grad_%d"), ("B", "gradB_%d")):
        e = 0.0
        for i, gi in enumerate(got):
            want = g[key % i]
            e = max(e, np.abs(gi - want).max() / max(np.abs(want).max(), 1e-300))
        errs[tag] = e
    which = check_branches_and_pick_reference(E, g, s) if "traj_stable" in g else None
    if which is not None:
        assert errs[which] < branch_tol, (which, errs, got)
    else:
        assert min(errs.values()) < tol, (errs, got)
    return which, errs


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




def _quat_to_mat(q):
    r, i, j, k = q
    s = 2.0 / (q * q).sum()
    return np.array([[1 - s * (j * j + k * k), s * (i * j - k * r), s * (i * k + j * r)],
                     [s * (i * j + k * r), 1 - s * (i * i + k * k), s * (j * k - i * r)],
                     [s * (i * k - j * r), s * (j * k + i * r), 1 - s * (i * i + j * j)]])


FACE_NORMAL1 = 1 << 30      # DSS_FACE_NORMAL1 (include/diffsdfsim_hip.h): the contact carries body 1's normal


def contact_branches(face, n):
    """Which body's normal each of the n contacts carries (1 = body 2's, the reference's `stable_mask`; 0 = body 1's): the flag
    the narrow phase stores in the contact's face word (where the normals of the two bodies coincide -- flat on flat -- the
    choice cannot be read off the normal itself, yet it decides which body's SDF the gradient flows through)."""
    w = np.where(face[:n] < 0, -1 - face[:n], face[:n])
    return ((w & FACE_NORMAL1) == 0).astype(np.int8)


def check_branches_and_pick_reference(E, g, s=0, margin=1e-9):
    """Compare the normal choice of every contact of every recorded sub-step with the reference's (`traj_stable`, recorded by
    oracle/gen/contact_record.py).  Where the two Laplacians the reference compared differ by more than rounding noise the
    build MUST take the same branch (asserted).  Returns 'A' / 'B' if the build's choices equal those of the reference's first
    / second (jittered) run at EVERY contact, coin flips included -- its gradient is then comparable to that run's to 1e-5 --
    else None."""
    tp, tnc, tb, tg, tf = E.get("tp_pose"), E.get("tp_nc"), E.get("tp_body"), E.get("tp_geom"), E.get("tp_face")
    k = len(g["traj_t"]) - 1
    same = {"A": True, "B": "traj_stableB" in g}
    steps = [(j, tf[j, s], tb[j, s], tg[j, s], int(tnc[j, s])) for j in range(0, k + 1)]      # slot 0: the contacts found at construction
    steps.append((k + 1, E.get("c_face")[s], E.get("c_body")[s], E.get("c_geom")[s], int(E.get("nc")[s])))
    for j, face, body, geom, n in steps:
        if j == 0:
            if "init_stable" not in g:
                return None
            ref_n, gb, gg = len(g["init_body"]), g["init_body"], g["init_geom"]
            ref_st, ref_lap, ref_stB = g["init_stable"], g["init_lap"], g.get("init_stableB")
        else:
            ref_n, gb, gg = int(g["traj_nc"][j - 1]), g["traj_body"][j - 1], g["traj_geom"][j - 1]
            ref_st, ref_lap, ref_stB = g["traj_stable"][j - 1], g["traj_lap"][j - 1], (g["traj_stableB"][j - 1] if "traj_stableB" in g else None)
        if n != ref_n:
            same["A"] = same["B"] = False      # another contact set: neither recorded run was reproduced step for step
            continue
        if n == 0:
            continue
        mine = contact_branches(face, n)
        # pair the build's contacts with the reference's by contact point (the order inside a body pair may differ)
        gb, gg = gb[:n], gg[:n]
        for c in range(n):
            cand = [r for r in range(n) if tuple(gb[r]) == (int(body[0, c]), int(body[1, c])) and np.abs(gg[r, 3:6] - geom[3:6, c]).max() < 1e-6]
            if len(cand) != 1:
                continue
            r = cand[0]
            ref, lap = int(ref_st[r]), ref_lap[r]
            if ref < 0:
                same["A"] = same["B"] = False     # an unrecorded choice: the run cannot be identified
                continue
            if abs(lap[1] - lap[0]) > margin * max(1.0, lap.max()):
                assert mine[c] == ref, ("normal taken from the other body than in the reference", j, c, lap)
            same["A"] &= mine[c] == ref
            if same["B"]:
                refb = int(ref_stB[r]) if ref_stB is not None else -1
                same["B"] &= refb >= 0 and mine[c] == refb
    return "A" if same["A"] else ("B" if same["B"] else None)


def force_reference_branches(E, g, s=0, run="A"):
    """Impose the reference's recorded normal choices (`traj_stable` / `init_stable` of run A, `*_stableB` of run B) on scene
    `s` before the reverse sweep: bit 30 of every taped contact's face word (`tp_face`, and `c_face` for the contacts found
    after the last sub-step) is overwritten with the reference's `stable_mask` (contacts.py:198) for the contact at the same
    point of the same body pair.  The reverse sweep reads the decision from there (step_bwd.hip) instead of repeating the
    twelve Laplacian probes, so the gradient it produces is the one of the reference run whose coin flips were imposed and
    can be held to north_star's 1e-5.  Where the two Laplacians differ by more than rounding noise the build's own decision
    must already equal the reference's (asserted).  Returns (contacts matched, bits flipped)."""
    sfx = "" if run == "A" else "B"
    tnc, tb, tg = E.get("tp_nc"), E.get("tp_body"), E.get("tp_geom")
    tf = E.get("tp_face")
    cf = E.get("c_face")
    k = len(g["traj_t"]) - 1
    matched = flipped = 0
    for j in range(0, k + 2):
        if j <= k:
            face, body, geom, n = tf[j, s], tb[j, s], tg[j, s], int(tnc[j, s])
        else:
            face, body, geom, n = cf[s], E.get("c_body")[s], E.get("c_geom")[s], int(E.get("nc")[s])
        if j == 0:
            gb, gg, ref_st, ref_lap = g["init_body"], g["init_geom"], g["init_stable" + sfx], g["init_lap"]
            ref_n = len(gb)
        else:
            gb, gg, ref_st, ref_lap = g["traj_body"][j - 1], g["traj_geom"][j - 1], g["traj_stable" + sfx][j - 1], g["traj_lap"][j - 1]
            ref_n = int(g["traj_nc"][j - 1])
        assert n == ref_n, ("contact count differs from the reference's", j, n, ref_n)
        used = set()
        for c in range(n):
            cand = [r for r in range(n) if r not in used and tuple(gb[r]) == (int(body[0, c]), int(body[1, c])) and np.abs(gg[r, 3:6] - geom[3:6, c]).max() < 1e-6]
            assert len(cand) >= 1, ("no reference contact at this point", j, c)
            # coincident contact points of one pair (shared mesh vertices) are matched one to one, in order: they carry equal
            # multipliers, so only the multiset of their decisions matters
            r = cand[0]
            used.add(r)
            ref = int(ref_st[r])
            if ref < 0:
                continue      # contacts the reference computed without gradients (the dt/2^10 escape): no decision to impose
            w = int(face[c]); neg = w < 0
            if neg:
                w = -1 - w
            mine = 0 if (w & FACE_NORMAL1) else 1
            lap = ref_lap[r]
            if sfx == "" and abs(lap[1] - lap[0]) > 1e-9 * max(1.0, lap.max()):
                assert mine == ref, ("normal taken from the other body than in the reference", j, c, lap)
            matched += 1
            if mine != ref:
                flipped += 1
                w = (w & ~FACE_NORMAL1) | (0 if ref else FACE_NORMAL1)
                face[c] = -1 - w if neg else w
    E.arr["tp_face"][:, s] = E.be.from_numpy(np.ascontiguousarray(tf[:, s]))
    E.arr["c_face"][s] = E.be.from_numpy(np.ascontiguousarray(cf[s]))
    return matched, flipped


def sweep(E):
    """Reverse sweep of d sum|pos_T|^2 over everything on the tape (the second half of rollout_and_sweep)."""
    adj = E._adjoint()
    ap = np.zeros_like(E.get("pose"))
    ap[:, :, 4:] = 2 * E.get("pose")[:, :, 4:]
    adj["a_pose"][...] = E.be.from_numpy(ap)
    adj["cur_slot"][...] = E.be.from_numpy((E.get("nsub") - 1).astype(np.int32))
    adj["lo_slot"][...] = -1
    E.backward_sweep(int(E.get("nsub").max()) + 1)


def grad_error(E, g, s=0, run="A"):
    got = param_grads(E, g, s)
    key = "grad_%d" if run == "A" else "gradB_%d"
    e = 0.0
    scale = max(np.abs(g[key % i]).max() for i in range(len(got)))
    for i, gi in enumerate(got):
        e = max(e, np.abs(gi - g[key % i]).max() / max(scale, 1e-300))
    return e


# ---- the benchmark's own scenes (tests/golden/bench_stack_s<k>.npz, bench_sphere_s<k>.npz; oracle/gen/gen_bench_golden.py) ----
BENCH_SEED = 1000      # bench.py steps scenes.box_stack(1024, seed=1000 + rank) / sphere_drop(256, seed=1000 + rank)


_BENCH_CACHE = {}


def bench_spec(kind, B, nref):
    """(cached per (kind, B, nref): callers must not modify the spec)"""
    key = (kind, B, nref)
    if key not in _BENCH_CACHE:
        _BENCH_CACHE[key] = _bench_spec(kind, B, nref)
    return _BENCH_CACHE[key]


def _bench_spec(kind, B, nref):
    """`scenes.box_stack(B, seed=1000)` / `scenes.sphere_drop(B, seed=1000)` -- the batch bench.py steps on rank 0 -- with the
    meshes of scenes 0..nref-1 and of the shared floor replaced by the reference's own (identical up to the last bit of interior
    grid coordinates: torch.linspace vs numpy), so that those scenes can be held against the goldens recorded from the
    reference.  Returns (spec, [golden of scene 0, ...])."""
    from diffsdfsim_amd import meshes, scenes
    spec = (scenes.box_stack if kind == "stack" else scenes.sphere_drop)(B, seed=BENCH_SEED)
    gs = [load_rollout("bench_%s_s%d" % (kind, s)) for s in range(nref)]
    g0 = gs[0]
    fd = spec["shape_prm"][0, 0]
    v, f, _tie = meshes.box_mesh(fd, grid_axes=(g0["floor_axes_w"], g0["floor_axes_h"], g0["floor_axes_d"]))
    assert (len(v), len(f)) == tuple(g0["meshsize_0"]) and np.abs(v - spec["meshes"][0][0]).max() < 1e-14
    spec["meshes"][0] = (v, f)
    for s, g in enumerate(gs):
        assert np.array_equal(g["pose0"], spec["pose"][s]) and np.array_equal(g["vel0"], spec["vel"][s]) and np.array_equal(g["shape_prm"], spec["shape_prm"][s])
        for b in range(1, spec["pose"].shape[1]):
            m = int(spec["mesh_id"][s, b])
            assert np.abs(g["verts_%d" % b] - spec["meshes"][m][0]).max() < 1e-14 and np.array_equal(g["faces_%d" % b], spec["meshes"][m][1])
            spec["meshes"][m] = (g["verts_%d" % b], g["faces_%d" % b])
    return spec, gs


def impose_reference_normals(E, g, s=0, run="A"):
    """Between two outer steps: give the CURRENT contacts of scene `s` the normal choice the reference made for them
    (`stable_mask`, contacts.py:198) -- flag AND normal.  The two candidate normals of a flat-on-flat contact (R2 n2 and -R1 n1)
    differ by the tilt between the two faces, so the choice is felt by the next LCP, i.e. by the forward trajectory, not only
    by the reverse sweep; a run that is to reproduce the reference's gradient to 1e-5 has to take the reference's side of
    every coin flip while it steps.  The normal written is the one recorded in the golden for the same contact point (it is
    what the kernel computes on that branch, to rounding).  Returns the number of contacts whose choice was changed."""
    sfx = "" if run == "A" else "B"
    j = int(E.get("nsub")[s])
    if j == 0:
        gb, gg, ref_st = g["init_body"], g["init_geom"], g["init_stable" + sfx]
        ref_n = len(gb)
    else:
        gb, gg, ref_st, ref_n = g["traj_body"][j - 1], g["traj_geom"][j - 1], g["traj_stable" + sfx][j - 1], int(g["traj_nc"][j - 1])
    n = int(E.get("nc")[s])
    assert n == ref_n, ("contact count differs from the reference's", j, n, ref_n)
    face, body, geom = E.get("c_face")[s], E.get("c_body")[s], E.get("c_geom")[s]
    used, changed = set(), 0
    for c in range(n):
        cand = [r for r in range(n) if r not in used and tuple(gb[r]) == (int(body[0, c]), int(body[1, c])) and np.abs(gg[r, 3:6] - geom[3:6, c]).max() < 1e-6]
        assert cand, ("no reference contact at this point", j, c)
        r = cand[0]; used.add(r)
        ref = int(ref_st[r])
        w = int(face[c])
        if ref < 0 or w < 0:
            continue
        mine = 0 if (w & FACE_NORMAL1) else 1
        if mine != ref:
            changed += 1
            face[c] = (w & ~FACE_NORMAL1) | (0 if ref else FACE_NORMAL1)
            geom[0:3, c] = gg[r, 0:3]
    if changed:
        E.arr["c_face"][s] = E.be.from_numpy(np.ascontiguousarray(face))
        E.arr["c_geom"][s] = E.be.from_numpy(np.ascontiguousarray(geom))
    return changed
