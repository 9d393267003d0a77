"""CPU restatement of the reference's 2-D analytic contact handler -- TEST INFRASTRUCTURE ONLY (tests/, smoke).

Follows `DiffContactHandler.__call__` (lcp_physics/physics/contacts.py:55-215) step by step in plain Python / numpy,
values only: circle-circle (:73-84), circle-polygon by the GJK walk (:85-120, with get_closest :298-332 and
get_barycentric_coords :334-357; the walk starts at vertex 0 where the reference draws a random vertex, :92) or, for a centre
inside the polygon, the separating-axis loop (:121-141), polygon-polygon by test_separations (:229-258), get_incident_edge
(:260-274) and clip_segment_to_line (:276-296).  Pinned by tests/golden/contacts2d.npz, which the reference itself produced
(oracle/gen/gen_contacts2d_golden.py); used to check the device kernels (csrc/contacts2d.hip, which find the closest feature
without a walk) on fresh random pairs where no golden exists.
"""
import numpy as np


def _left(v):                                    # utils.py:124-127
    return np.array([v[1], -v[0]])


def _bary2(pt, a, b):                            # contacts.py:335-341
    diff = b - a
    n = np.linalg.norm(diff)
    nd = diff / n
    return np.dot(b - pt, nd) / n, np.dot(pt - a, nd) / n


def _bary3(pt, s):                               # contacts.py:342-352
    M = np.array([[s[0][0], s[1][0], s[2][0]], [s[0][1], s[1][1], s[2][1]], [1.0, 1.0, 1.0]])
    return np.linalg.solve(M, np.array([pt[0], pt[1], 1.0]))


def _closest(pt, s):                             # contacts.py:298-332
    if len(s) == 1:
        return s[0], [0]
    if len(s) == 2:
        u, v = _bary2(pt, s[0], s[1])
        if u <= 0:
            return s[1], [1]
        if v <= 0:
            return s[0], [0]
        return u * s[0] + v * s[1], [0, 1]
    uAB, vAB = _bary2(pt, s[0], s[1])
    uBC, vBC = _bary2(pt, s[1], s[2])
    uCA, vCA = _bary2(pt, s[2], s[0])
    uABC, vABC, wABC = _bary3(pt, s)
    if vAB <= 0 and uCA <= 0:
        return s[0], [0]
    if vBC <= 0 and uAB <= 0:
        return s[1], [1]
    if vCA <= 0 and uBC <= 0:
        return s[2], [2]
    if uAB > 0 and vAB > 0 and wABC <= 0:
        return uAB * s[0] + vAB * s[1], [0, 1]
    if uBC > 0 and vBC > 0 and uABC <= 0:
        return uBC * s[1] + vBC * s[2], [1, 2]
    if uCA > 0 and vCA > 0 and vABC <= 0:
        return uCA * s[2] + vCA * s[0], [2, 0]
    return pt, [0, 1, 2]


def _support(verts, d):                          # contacts.py:218-227 (last of the largest; none below -1)
    best, idx = -1.0, -1
    for i, p in enumerate(verts):
        c = float(np.dot(p, d))
        if c >= best:
            best, idx = c, i
    return idx


def _circle_polygon(cpos, crad, hpos, hv, sat, eps):
    """-> (contact or None, sat); contact = (normal, pt1 (from the circle), pt2 (from the polygon), dist)"""
    test = cpos - hpos
    simplex, idx = [hv[0]], [0]
    while True:
        closest, used = _closest(test, simplex)
        if len(used) == 3:
            break
        if len(used) == 2:
            sd = _left(simplex[used[0]] - simplex[used[1]])
            if np.dot(sd, test - simplex[used[0]]) < 0:
                sd = -sd
        else:
            sd = test - closest
        if sd[0] == 0 and sd[1] == 0:
            break
        si = _support(hv, sd)
        if si in idx:
            break
        simplex = [simplex[k] for k in used] + [hv[si]]
        idx = [idx[k] for k in used] + [si]
    if len(used) < 3:
        pt2 = closest
        pt1 = closest + hpos - cpos
        dist = np.linalg.norm(pt1) - crad
        if dist > eps:
            return None, sat
        return (-pt1 / np.linalg.norm(pt1), pt1, pt2, dist), sat
    best, res, n = -1e10, None, len(hv)
    for i in range(sat, n + sat):
        k = i % n
        edge = hv[(k + 1) % n] - hv[k]
        nrm = _left(edge) / np.linalg.norm(edge)
        dist = np.dot(nrm, test - hv[k]) - crad
        if dist > best:
            sat = k
            if dist > eps:
                return None, sat
            best = dist
            pt2 = test + nrm * -(dist + crad)
            res = (nrm, pt2 + hpos - cpos, pt2, dist)
    return res, sat


def _separations(p1, v1, p2, v2, start, eps):    # contacts.py:229-258
    n, best, out = len(v1), -1e10, None
    for i in range(start, n + start):
        k = i % n
        edge = v1[(k + 1) % n] - v1[k]
        en = np.linalg.norm(edge)
        nrm = _left(edge) / en
        si = _support(v2, -nrm)
        sp = v2[si] + p2 - p1
        dist = np.dot(nrm, sp - v1[k])
        if dist > best:
            if dist > eps:
                return dist, None, k
            best = dist
            out = (nrm, si, en, k)
    return best, out, out[3]


def _incident_edge(nrm, verts, iv):              # contacts.py:260-274
    n, md, be = len(verts), 1e10, -1
    for i in ((iv - 1) % n, iv):
        edge = verts[(i + 1) % n] - verts[i]
        d = float(np.dot(nrm, _left(edge) / np.linalg.norm(edge)))
        if d < md:
            md, be = d, i
    return be


def _clip(verts, nrm, off):                      # contacts.py:276-296
    out = []
    d0, d1 = np.dot(nrm, verts[0]) + off, np.dot(nrm, verts[1]) + off
    if d0 >= 0.0:
        out.append(verts[0])
    if d1 >= 0.0:
        out.append(verts[1])
    if d0 * d1 < 0.0 or len(out) < 2:
        out.append(verts[0] + d0 / (d0 - d1) * (verts[1] - verts[0]))
    return out


def pair(kind, pos, rad, verts, sat, eps):
    """One pair.  kind [2], pos [2][2], rad [2], verts: two lists of vertices, sat [2] -> (list of (n, p1, p2, pen), sat)."""
    sat = list(sat)
    if kind[0] == 0 and kind[1] == 0:
        r = rad[0] + rad[1]
        d = pos[0] - pos[1]
        dist = np.linalg.norm(d)
        pen = r - dist
        if pen < -eps:
            return [], sat
        n = d / dist
        return [(n, -n * (rad[0] - pen / 2), n * (rad[1] - pen / 2), pen)], sat
    if kind[0] == 0 or kind[1] == 0:
        c, h = (0, 1) if kind[0] == 0 else (1, 0)
        res, sat[h] = _circle_polygon(pos[c], rad[c], pos[h], verts[h], sat[h], eps)
        if res is None:
            return [], sat
        n, pt1, pt2, dist = res
        if c == 1:
            n, pt1, pt2 = -n, pt2, pt1
        return [(n, pt1, pt2, -dist)], sat
    d1, c1, sat[0] = _separations(pos[0], verts[0], pos[1], verts[1], sat[0], eps)
    if d1 > eps:
        return [], sat
    d2, c2, sat[1] = _separations(pos[1], verts[1], pos[0], verts[0], sat[1], eps)
    if d2 > eps:
        return [], sat
    ref, inc = (1, 0) if d2 > d1 else (0, 1)
    nrm, iv, en, re = c2 if ref == 1 else c1
    ie = _incident_edge(nrm, verts[inc], iv)
    nv = len(verts[inc])
    seg = [verts[inc][ie] + pos[inc] - pos[ref], verts[inc][(ie + 1) % nv] + pos[inc] - pos[ref]]
    cp = _left(nrm)
    cl = _clip(seg, cp, en / 2)
    if len(cl) < 2:
        return [], sat
    cl = _clip(cl, -cp, en / 2)
    out = []
    for v in cl:
        dist = np.dot(nrm, v - verts[ref][re])
        if dist <= eps:
            on_ref = v + nrm * -dist
            on_inc = on_ref + pos[ref] - pos[inc]
            out.append((nrm, on_inc, on_ref, -dist) if ref == 1 else (-nrm, on_ref, on_inc, -dist))
    return out, sat


def contacts2d(kind, nv, pos, rad, verts, sat_in, eps):
    """The batch layout of dss_contacts2d_forward: kind / nv / sat_in [2][P], pos [2][P][2], rad [2][P], verts [2][P][maxv][2]
    -> out [P][2][7], count [P], sat_out [2][P]."""
    P = pos.shape[1]
    out, count, sat_out = np.zeros((P, 2, 7)), np.zeros(P, np.int32), np.array(sat_in, np.int32).copy()
    for p in range(P):
        vs = [[verts[s, p, i].astype(float) for i in range(int(nv[s, p]))] for s in range(2)]
        cs, sat = pair(kind[:, p], [pos[0, p].astype(float), pos[1, p].astype(float)], rad[:, p], vs, sat_in[:, p], eps)
        count[p] = len(cs)
        sat_out[:, p] = sat
        for q, (n, p1, p2, pen) in enumerate(cs):
            out[p, q] = np.concatenate([n, p1, p2, [pen]])
    return out, count, sat_out


def random_pairs(rng, P, maxv=8):
    """Random near-touching pairs in the batch layout (test input generator): circles of radius 0.3-1, convex polygons of
    3-7 vertices in the reference's orientation (clockwise in its y-down frame, about the centroid)."""
    kind = rng.integers(0, 2, (2, P)).astype(np.int32)
    nv = np.zeros((2, P), np.int32); pos = np.zeros((2, P, 2)); rad = np.zeros((2, P)); verts = np.zeros((2, P, maxv, 2))
    sat = np.zeros((2, P), np.int32)
    for p in range(P):
        ext = []
        th = rng.uniform(0, 2 * np.pi)
        d = np.array([np.cos(th), np.sin(th)])
        for s in range(2):
            if kind[s, p] == 0:
                rad[s, p] = rng.uniform(0.3, 1.0)
                ext.append(rad[s, p])
                continue
            n = int(rng.integers(3, 8))
            while True:
                ang = np.sort(rng.uniform(0, 2 * np.pi, n))
                if np.min(np.diff(np.concatenate([ang, [ang[0] + 2 * np.pi]]))) > 0.3:
                    break
            size = rng.uniform(0.5, 1.2)
            v = np.stack([size * rng.uniform(0.6, 1.0) * np.cos(ang), size * rng.uniform(0.6, 1.0) * np.sin(ang)], 1)
            tot = sum((v[(i + 1) % n][0] - v[i][0]) * (v[(i + 1) % n][1] + v[i][1]) for i in range(n))
            if tot >= 0:
                v = v[::-1].copy()
            # centroid of the polygon (bodies.py:243-254), vertices relative to it
            num, den = np.zeros(2), 0.0
            for i in range(n):
                a, b = v[i], v[(i + 1) % n]
                cr = b[0] * a[1] - b[1] * a[0]
                num += cr * (a + b); den += cr / 2
            v = v - num / (6 * den)
            nv[s, p] = n; verts[s, p, :n] = v; sat[s, p] = rng.integers(0, n)
            ext.append(max(float(np.dot(q, d if s == 1 else -d)) for q in v))
        gap = rng.uniform(-0.25, 0.3) if rng.uniform() < 0.8 else rng.uniform(-1.5, -0.3)
        pos[1, p] = rng.uniform(-1, 1, 2)
        pos[0, p] = pos[1, p] + max(ext[0] + ext[1] + gap, 0.02) * d
    return dict(kind=kind, nv=nv, pos=pos, rad=rad, verts=verts, sat_in=sat)
