// contact_geom.h -- contact point geometry shared by the narrow phase (T = double) and its
// backward (T = Dual<N>).  Restates FWContactHandler._compute_contacts
// (sdf_physics/physics3d/contacts.py:161-214) for SDF-SDF pairs.
#pragma once
#include "geom.h"

namespace dss {
#if DSS_ALL_SHAPES
inline namespace shapes_all {
#else
inline namespace shapes_lean {
#endif

template <class T> struct BodyG {
    T q[4], pos[3];
    Shape<T> shape;
};

// Laplacian probe of phi by central second differences with step h (values only; the reference
// uses it in a comparison, contacts.py:184-198, so it carries no gradient).
template <class T> __host__ __device__ inline double lap_probe(const Shape<T> &s, const T *p, double phi0, double h)
{
    Shape<double> sd;
    sd.type = s.type;
    for (int i = 0; i < 3; ++i) { sd.prm[i] = val(s.prm[i]); sd.hd[i] = val(s.hd[i]); }
    sd.scale = val(s.scale);
#if DSS_ALL_SHAPES
    sd.hr = val(s.hr);
    sd.lin = nullptr;
    sd.grid = s.grid;
    for (int i = 0; i < 3; ++i) sd.gn[i] = s.gn[i];
#endif
    double acc = 0.0, pt[3] = {val(p[0]), val(p[1]), val(p[2])}, g[3];
    for (int i = 0; i < 3; ++i) {
        double a, b, save = pt[i];
        pt[i] = save + h; query_sdf(sd, pt, a, g, false);
        pt[i] = save - h; query_sdf(sd, pt, b, g, false);
        pt[i] = save;
        acc += a - 2.0 * phi0 + b;
    }
    return acc;
}

// One contact from a barycentric point on a triangle of body 1's mesh, in two halves (the backward differentiates
// the second half alone for the inputs that only enter there).
//   head: body-1 side.  tri: the three vertices (body-1 frame), abc: barycentrics (constants)
//         -> cp1 (surface point, body-1 frame), n1 (normal there, body-1 frame), d1, p1 = R1 cp1 (world offset)
//   tail: body-2 side.  -> n (world), p2 (world-frame offset from body 2's origin), pen = -phi_2
//   stable_io: < 0 on entry = decide which body's normal is used from the two Laplacian probes and report it (0/1);
//              >= 0 = use that decision (the backward's derivative passes re-use the one of their value pass: the
//              probes carry no gradient and cost twelve SDF evaluations)
template <class T>
__host__ __device__ inline void contact_head(const BodyG<T> &b1, const T tri[3][3], const double abc[3], T *cp1, T *n1, T &d1,
                                             T *p1)
{
    for (int i = 0; i < 3; ++i) cp1[i] = tri[0][i] * abc[0] + tri[1][i] * abc[1] + tri[2][i] * abc[2];
    // the triangle point is pulled onto body 1's true surface by one Newton step (contacts.py:169-171)
#if DSS_ALL_SHAPES
    if (b1.shape.type == SHAPE_IGR) {      // (reverse sweep only: records 0 and 1 of the body's queries for this contact)
        igr_lin(b1.shape, 0, cp1, d1, n1);
        for (int i = 0; i < 3; ++i) cp1[i] = cp1[i] - d1 * n1[i];
        igr_lin(b1.shape, 1, cp1, d1, n1);
        quat_apply(b1.q, cp1, p1);
        return;
    }
#endif
    query_sdf(b1.shape, cp1, d1, n1, true);
    for (int i = 0; i < 3; ++i) cp1[i] = cp1[i] - d1 * n1[i];
    query_sdf(b1.shape, cp1, d1, n1, true);
    quat_apply(b1.q, cp1, p1);
}
template <class T>
__host__ __device__ inline void contact_tail(const BodyG<T> &b1, const BodyG<T> &b2, const T *cp1, const T *n1, const T &d1,
                                             const T *p1, double lap_h, T *n, T *p2, T &pen, int *stable_io, bool detach_b2 = false)
{
    T cpw[3], rel[3], cp2[3], d2, n2[3], t[3];
    for (int i = 0; i < 3; ++i) { cpw[i] = p1[i] + b1.pos[i]; rel[i] = cpw[i] - b2.pos[i]; }
    quat_apply_inv(b2.q, rel, cp2);
    if (detach_b2) for (int i = 0; i < 3; ++i) cp2[i] = T(val(cp2[i]));     // World3D(detach_contact_b2=True), contacts.py:175-178
#if DSS_ALL_SHAPES
    if (b2.shape.type == SHAPE_IGR) igr_lin(b2.shape, 0, cp2, d2, n2);
    else
#endif
    query_sdf(b2.shape, cp2, d2, n2, true);
    bool stable;
    if (stable_io && *stable_io >= 0) stable = *stable_io != 0;
    else {
        const double l1 = lap_probe(b1.shape, cp1, val(d1), lap_h), l2 = lap_probe(b2.shape, cp2, val(d2), lap_h);
        stable = fabs(l2) < fabs(l1);
        if (stable_io) *stable_io = stable ? 1 : 0;
    }
    if (stable) {
        quat_apply(b2.q, n2, n);
    } else {
        quat_apply(b1.q, n1, t);
        for (int i = 0; i < 3; ++i) n[i] = -t[i];
    }
    for (int i = 0; i < 3; ++i) t[i] = cp2[i] - d2 * n2[i];
    quat_apply(b2.q, t, p2);
    pen = -d2;
}
template <class T>
__host__ __device__ inline void contact_from_bary(const BodyG<T> &b1, const BodyG<T> &b2, const T tri[3][3],
                                                  const double abc[3], double lap_h, T *n, T *p1, T *p2, T &pen,
                                                  int *stable_io = nullptr, bool detach_b2 = false)
{
    T cp1[3], n1[3], d1;
    contact_head(b1, tri, abc, cp1, n1, d1, p1);
    contact_tail(b1, b2, cp1, n1, d1, p1, lap_h, n, p2, pen, stable_io, detach_b2);
}

// ---- time-of-contact distance function (World.H.D, lcp_physics/physics/world.py:150-174) composed with the
// preprocessing of its arguments at world.py:276-319.  Input order (43): h, hc (the dt_ used to rewind the poses),
// p1(3) p2(3) n(3) of the new contact, V1(6) V2(6) new velocities, q1(4) x1(3) q2(4) x2(3) poses after the move,
// a1(3) a2(3) linear accelerations f/m.
template <class T> __host__ __device__ inline T toc_D(const T *in)
{
    const T h = in[0], hc = in[1];
    const T *p1 = in + 2, *p2 = in + 5, *n = in + 8, *V1 = in + 11, *V2 = in + 17, *q1 = in + 23, *x1 = in + 27,
            *q2 = in + 30, *x2 = in + 34, *a1 = in + 37, *a2 = in + 40;
    T R1[9], R2[9], r1[9], r2[9], R10[9], R20[9], w[3], pos1[3], pos2[3];
    for (int i = 0; i < 3; ++i) { pos1[i] = x1[i] - hc * V1[3 + i]; pos2[i] = x2[i] - hc * V2[3 + i]; }
    for (int i = 0; i < 3; ++i) w[i] = -hc * V1[i];
    so3_exp(w, r1);
    for (int i = 0; i < 3; ++i) w[i] = -hc * V2[i];
    so3_exp(w, r2);
    quat_to_mat(q1, R1);
    quat_to_mat(q2, R2);
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            R10[3 * a + b] = r1[3 * a] * R1[b] + r1[3 * a + 1] * R1[3 + b] + r1[3 * a + 2] * R1[6 + b];
            R20[3 * a + b] = r2[3 * a] * R2[b] + r2[3 * a + 1] * R2[3 + b] + r2[3 * a + 2] * R2[6 + b];
        }
    T cs1[3], cs2[3], ns2[3];
    for (int i = 0; i < 3; ++i) {   // R0^T v
        cs1[i] = R10[i] * p1[0] + R10[3 + i] * p1[1] + R10[6 + i] * p1[2];
        cs2[i] = R20[i] * p2[0] + R20[3 + i] * p2[1] + R20[6 + i] * p2[2];
        ns2[i] = R20[i] * n[0] + R20[3 + i] * n[1] + R20[6 + i] * n[2];
    }
    T dRi[9], dRj[9], Rih[9], Rjh[9];
    for (int i = 0; i < 3; ++i) w[i] = h * V1[i];
    so3_exp(w, dRi);
    for (int i = 0; i < 3; ++i) w[i] = h * V2[i];
    so3_exp(w, dRj);
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            Rih[3 * a + b] = dRi[3 * a] * R10[b] + dRi[3 * a + 1] * R10[3 + b] + dRi[3 * a + 2] * R10[6 + b];
            Rjh[3 * a + b] = dRj[3 * a] * R20[b] + dRj[3 * a + 1] * R20[3 + b] + dRj[3 * a + 2] * R20[6 + b];
        }
    T ciw[3], rel[3], D = T(0.0);
    for (int i = 0; i < 3; ++i) {
        const T pih = pos1[i] + h * V1[3 + i] + 0.5 * a1[i] * h * h, pjh = pos2[i] + h * V2[3 + i] + 0.5 * a2[i] * h * h;
        ciw[i] = Rih[3 * i] * cs1[0] + Rih[3 * i + 1] * cs1[1] + Rih[3 * i + 2] * cs1[2] + pih;
        rel[i] = ciw[i] - pjh;
    }
    for (int i = 0; i < 3; ++i) {
        const T cij = Rjh[i] * rel[0] + Rjh[3 + i] * rel[1] + Rjh[6 + i] * rel[2];   // Rjh^T rel
        D = D + ns2[i] * (cs2[i] - cij);
    }
    return D;
}

// friction directions of World3D.Jf (physics3d/world.py:84-94; utils.py:247-256 `orthogonal`)
template <class T> __host__ __device__ inline void friction_dirs(const T *n, int nd, T D[][3])
{
    int am = 0;
    for (int i = 1; i < 3; ++i) if (fabs(val(n[i])) < fabs(val(n[am]))) am = i;
    T e[3] = {T(am == 0 ? 1.0 : 0.0), T(am == 1 ? 1.0 : 0.0), T(am == 2 ? 1.0 : 0.0)}, t[3];
    cross(e, n, t);
    normalize(t, D[0]);
    cross(D[0], n, t);
    normalize(t, D[1]);
    if (nd == 4) {
        for (int i = 0; i < 3; ++i) t[i] = D[0][i] + D[1][i];
        normalize(t, D[2]);
        cross(D[2], n, t);
        normalize(t, D[3]);
    }
}

}  // inline namespace shapes_*
}  // namespace dss
