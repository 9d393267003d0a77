// contact_rev.h -- REVERSE-mode adjoint of one contact's geometry (contact_geom.h: contact_head + contact_tail) for the lean
// kernel variant (box / sphere / cylinder bodies), restating what torch.autograd does with FWContactHandler._compute_contacts
// (sdf_physics/physics3d/contacts.py:161-214) and SDF3D.query_sdfs (bodies.py:721-760).
//
// The forward-mode version (step_bwd.hip: contact_vjp, Dual<4>) differentiates the same chain five times over, once per
// group of inputs; here the chain is walked forward once in plain doubles and backward once, every primitive handing its
// output adjoint to its inputs.  The non-smooth pieces follow geom.h's Dual rules, which follow torch's: abs'(0) = 0,
// clamp passes the gradient on its boundary, max(a, b) routes it to one argument (the first on a tie), torch.max(q, 0)
// splits a tie, sqrt'(0) = 0, F.normalize clamps the norm at 1e-12 (a clamped norm is a constant).
#pragma once
#include "contact_geom.h"

namespace dss {
inline namespace shapes_lean {

// o = quaternion_apply(q, p) = (q (0,p) q*)[1:]  ->  qbar += , pbar +=   (<x, a b> = <x b*, a> = <a* x, b> for the Hamilton product)
__host__ __device__ inline void rev_quat_apply(const double *q, const double *p, const double *obar, double *qbar, double *pbar)
{
    const double pq[4] = {0.0, p[0], p[1], p[2]}, qi[4] = {q[0], -q[1], -q[2], -q[3]};
    double t[4];
    quat_raw_mul(q, pq, t);
    const double rbar[4] = {0.0, obar[0], obar[1], obar[2]};
    // r = t qi:  tbar = rbar qi* = rbar q ;  qibar = t* rbar
    double tbar[4], qibar[4];
    quat_raw_mul(rbar, q, tbar);
    const double tc[4] = {t[0], -t[1], -t[2], -t[3]};
    quat_raw_mul(tc, rbar, qibar);
    // t = q pq:  qbar += tbar pq* ;  pqbar = q* tbar
    const double pqc[4] = {0.0, -p[0], -p[1], -p[2]};
    double a[4], b[4];
    quat_raw_mul(tbar, pqc, a);
    quat_raw_mul(qi, tbar, b);
    qbar[0] += a[0] + qibar[0];
    for (int i = 1; i < 4; ++i) qbar[i] += a[i] - qibar[i];     // qi = conj(q)
    for (int i = 0; i < 3; ++i) pbar[i] += b[1 + i];
}

// y = F.normalize(a) = a / max(|a|, 1e-12)  ->  abar +=
__host__ __device__ inline void rev_normalize(const double *a, const double *ybar, double *abar)
{
    const double s = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    const double n = sqrt(s);
    if (n < 1e-12) { for (int i = 0; i < 3; ++i) abar[i] += ybar[i] / 1e-12; return; }
    const double y[3] = {a[0] / n, a[1] / n, a[2] / n};
    const double d = y[0] * ybar[0] + y[1] * ybar[1] + y[2] * ybar[2];
    for (int i = 0; i < 3; ++i) abar[i] += (ybar[i] - y[i] * d) / n;
}
__host__ __device__ inline void fwd_normalize(const double *a, double *y)
{
    double n = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (n < 1e-12) n = 1e-12;
    for (int i = 0; i < 3; ++i) y[i] = a[i] / n;
}
__host__ __device__ inline double sgn0(double x) { return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : 0.0); }

// adjoint of the shape constants a body's queries read (make_shape): scale, hd[3]
struct ShapeBar { double scale, hd[3]; };

// (u, g) = sdf_unit(shape, p) with the gradient  ->  pbar += , sb.hd +=      (geom.h: sdf_unit, want_grad = true)
__host__ __device__ inline void rev_sdf_unit(const Shape<double> &s, const double *p, double ubar, const double *gbar, double *pbar,
                                             ShapeBar &sb)
{
    if (s.type == SHAPE_BOX) {
        double q[3], m[3], mg[3], nmv[3], go[3], g1[3], sg[3], mdd[3];
        for (int i = 0; i < 3; ++i) q[i] = fabs(p[i]) - s.hd[i];
        int im = q[1] > q[0] ? 1 : 0;
        if (q[2] > q[im]) im = 2;
        const double md = q[im];
        for (int i = 0; i < 3; ++i) { m[i] = q[i] >= 0.0 ? q[i] : 0.0; mg[i] = q[i] > 0.0 ? q[i] : 0.0; }
        const double r = sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
        fwd_normalize(mg, nmv);
        for (int i = 0; i < 3; ++i) {
            sg[i] = p[i] < 0.0 ? -1.0 : 1.0;
            mdd[i] = (md <= 0.0 && q[i] == md) ? 1.0 : 0.0;
            go[i] = (nmv[i] + mdd[i]) * sg[i];
        }
        fwd_normalize(go, g1);
        // ---- reverse
        double g1bar[3] = {0, 0, 0}, gobar[3] = {0, 0, 0}, nmvbar[3], mgbar[3] = {0, 0, 0}, qbar[3] = {0, 0, 0};
        rev_normalize(g1, gbar, g1bar);
        rev_normalize(go, g1bar, gobar);
        for (int i = 0; i < 3; ++i) nmvbar[i] = gobar[i] * sg[i];
        rev_normalize(mg, nmvbar, mgbar);
        for (int i = 0; i < 3; ++i) qbar[i] += mgbar[i] * (q[i] > 0.0 ? 1.0 : (q[i] == 0.0 ? 0.5 : 0.0));
        if (r > 0.0) for (int i = 0; i < 3; ++i) qbar[i] += (q[i] >= 0.0 ? ubar * m[i] / r : 0.0);
        if (md <= 0.0) qbar[im] += ubar;
        for (int i = 0; i < 3; ++i) { sb.hd[i] -= qbar[i]; pbar[i] += qbar[i] * sgn0(p[i]); }
    } else if (s.type == SHAPE_CYLINDER) {
        const double s2 = p[0] * p[0] + p[1] * p[1], rho = sqrt(s2);
        const double q0 = fabs(rho) - s.hd[0], q1 = fabs(p[2]) - s.hd[1];
        const int im = q1 > q0 ? 1 : 0;
        const double md = im ? q1 : q0;
        const double m0 = q0 >= 0.0 ? q0 : 0.0, m1 = q1 >= 0.0 ? q1 : 0.0;
        const double r = sqrt(m0 * m0 + m1 * m1);
        const bool nm_c = r < 1e-12, rxy_c = rho < 1e-12;
        const double nm = nm_c ? 1e-12 : r, rxy = rxy_c ? 1e-12 : rho;
        const double inside = md <= 0.0 ? 1.0 : 0.0;
        const double g0 = m0 / nm + inside * (q0 == md ? 1.0 : 0.0), g1s = m1 / nm + inside * (q1 == md ? 1.0 : 0.0);
        const double sg = p[2] < 0.0 ? -1.0 : 1.0;
        const double e0 = p[0] / rxy, e1 = p[1] / rxy;
        const double go[3] = {g0 * e0, g0 * e1, g1s * sg};
        double g1[3];
        fwd_normalize(go, g1);
        // ---- reverse
        double g1bar[3] = {0, 0, 0}, gobar[3] = {0, 0, 0};
        rev_normalize(g1, gbar, g1bar);
        rev_normalize(go, g1bar, gobar);
        const double g0bar = gobar[0] * e0 + gobar[1] * e1, g1sbar = gobar[2] * sg;
        const double e0bar = gobar[0] * g0, e1bar = gobar[1] * g0;
        double rxybar = -(e0bar * p[0] + e1bar * p[1]) / (rxy * rxy), rhobar = rxy_c ? 0.0 : rxybar;
        pbar[0] += e0bar / rxy; pbar[1] += e1bar / rxy;
        double m0bar = g0bar / nm, m1bar = g1sbar / nm, nmbar = -(g0bar * m0 + g1sbar * m1) / (nm * nm);
        // nm = sqrt(m0^2 + m1^2) (unless clamped) and the value's r: the same square root
        double rbar = ubar + (nm_c ? 0.0 : nmbar);
        if (r > 0.0) { m0bar += rbar * m0 / r; m1bar += rbar * m1 / r; }
        double q0bar = q0 >= 0.0 ? m0bar : 0.0, q1bar = q1 >= 0.0 ? m1bar : 0.0;
        if (md <= 0.0) { if (im) q1bar += ubar; else q0bar += ubar; }
        sb.hd[0] -= q0bar; sb.hd[1] -= q1bar;
        rhobar += q0bar * sgn0(rho);
        pbar[2] += q1bar * sgn0(p[2]);
        if (s2 > 0.0) { pbar[0] += rhobar * p[0] / rho; pbar[1] += rhobar * p[1] / rho; }
    } else {   // sphere
        const double s2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2], r = sqrt(s2);
        double g1[3];
        fwd_normalize(p, g1);
        double g1bar[3] = {0, 0, 0};
        rev_normalize(g1, gbar, g1bar);
        rev_normalize(p, g1bar, pbar);
        if (s2 > 0.0) for (int i = 0; i < 3; ++i) pbar[i] += ubar * p[i] / r;
        sb.hd[0] -= ubar;
    }
}

// (phi, g) = query_sdf(shape, pt)  ->  ptbar += , sb +=          (geom.h: query_sdf; phi and g recomputed by the caller)
__host__ __device__ inline void rev_query_sdf(const Shape<double> &s, const double *pt, double phibar, const double *gbar, double *ptbar,
                                              ShapeBar &sb)
{
    const double sc = s.scale;
    if (!(fabs(pt[0]) <= sc && fabs(pt[1]) <= sc && fabs(pt[2]) <= sc)) { sb.scale += phibar; return; }   // phi = scale, g = 0
    double p[3], u, g[3];
    for (int i = 0; i < 3; ++i) p[i] = pt[i] / sc;
    sdf_unit(s, p, u, g, false);
    double pbar[3] = {0, 0, 0};
    rev_sdf_unit(s, p, phibar * sc, gbar, pbar, sb);
    sb.scale += phibar * u;
    for (int i = 0; i < 3; ++i) { ptbar[i] += pbar[i] / sc; sb.scale -= pbar[i] * p[i] / sc; }
}

// make_shape (geom.h): (scale, hd) from the body's parameters  ->  prmbar +=
__host__ __device__ inline void rev_make_shape(int type, const double *prm, const Shape<double> &s, ShapeBar sb, double *prmbar)
{
    const double sc = s.scale;
    if (type == SHAPE_BOX) {
        for (int i = 0; i < 3; ++i) { prmbar[i] += sb.hd[i] / (2.0 * sc); sb.scale -= sb.hd[i] * prm[i] / (2.0 * sc * sc); }
        int im = prm[1] > prm[0] ? 1 : 0;
        if (prm[2] > prm[im]) im = 2;
        prmbar[im] += sb.scale * 1.5 / 2.0;
    } else if (type == SHAPE_CYLINDER) {
        prmbar[0] += sb.hd[0] / sc; sb.scale -= sb.hd[0] * prm[0] / (sc * sc);
        prmbar[1] += sb.hd[1] / (2.0 * sc); sb.scale -= sb.hd[1] * prm[1] / (2.0 * sc * sc);
        if (prm[1] / 2.0 > prm[0]) prmbar[1] += sb.scale * 1.5 / 2.0; else prmbar[0] += sb.scale * 1.5;
    } else {
        prmbar[0] += sb.hd[0] / sc; sb.scale -= sb.hd[0] * prm[0] / (sc * sc);
        prmbar[0] += sb.scale * 1.5;
    }
}

// d <gbar, (n, p1, p2)> / d (q1, x1, q2, x2, prm1, prm2) for one contact -> out[20] (the layout of contact_vjp)
//   tv / tg: the triangle's vertices (body-1 frame) and their derivatives w.r.t. the body's own parameter (engine.default_vgrad)
//   stable: which body's normal the forward pass took (1 = body 2's); detach_b2: World3D(detach_contact_b2=True)
__host__ __device__ inline void contact_vjp_rev(const double *P1, const double *P2, int ty1, int ty2, const double *prm1, const double *prm2,
                                                const double tv[3][3], const double tg[3][3], const double *abc, const double *gbar,
                                                int stable, bool detach_b2, double *out)
{
    Shape<double> S1, S2;
    make_shape(S1, ty1, prm1);
    make_shape(S2, ty2, prm2);
    const double *q1 = P1, *x1 = P1 + 4, *q2 = P2, *x2 = P2 + 4;
    // ---- forward, values only
    double c0[3], dA, nA[3], c1[3], d1, n1[3], p1[3], rel[3], c2[3], d2, n2[3], t[3];
    for (int i = 0; i < 3; ++i) c0[i] = tv[0][i] * abc[0] + tv[1][i] * abc[1] + tv[2][i] * abc[2];
    query_sdf(S1, c0, dA, nA, true);
    for (int i = 0; i < 3; ++i) c1[i] = c0[i] - dA * nA[i];
    query_sdf(S1, c1, d1, n1, true);
    quat_apply(q1, c1, p1);
    for (int i = 0; i < 3; ++i) rel[i] = (p1[i] + x1[i]) - x2[i];
    const double q2c[4] = {q2[0], -q2[1], -q2[2], -q2[3]};
    quat_apply(q2c, rel, c2);
    query_sdf(S2, c2, d2, n2, true);
    for (int i = 0; i < 3; ++i) t[i] = c2[i] - d2 * n2[i];
    // ---- reverse
    const double *nbar = gbar, *p1bar = gbar + 3, *p2bar = gbar + 6;
    double q1b[4] = {0, 0, 0, 0}, q2b[4] = {0, 0, 0, 0}, x1b[3] = {0, 0, 0}, x2b[3] = {0, 0, 0}, pr1b[3] = {0, 0, 0}, pr2b[3] = {0, 0, 0};
    ShapeBar sb1 = {0.0, {0, 0, 0}}, sb2 = {0.0, {0, 0, 0}};
    double tb[3] = {0, 0, 0}, c2b[3] = {0, 0, 0}, n2b[3] = {0, 0, 0}, n1b[3] = {0, 0, 0}, d2b = 0.0;
    rev_quat_apply(q2, t, p2bar, q2b, tb);                       // p2 = R2 (c2 - d2 n2)
    for (int i = 0; i < 3; ++i) { c2b[i] += tb[i]; d2b -= tb[i] * n2[i]; n2b[i] -= d2 * tb[i]; }
    if (stable) rev_quat_apply(q2, n2, nbar, q2b, n2b);          // n = R2 n2 | -R1 n1
    else { const double mn[3] = {-nbar[0], -nbar[1], -nbar[2]}; rev_quat_apply(q1, n1, mn, q1b, n1b); }
    rev_query_sdf(S2, c2, d2b, n2b, c2b, sb2);
    double p1tot[3] = {p1bar[0], p1bar[1], p1bar[2]};
    if (!detach_b2) {                                            // c2 = R2^T (p1 + x1 - x2)
        double qcb[4] = {0, 0, 0, 0}, relb[3] = {0, 0, 0};
        rev_quat_apply(q2c, rel, c2b, qcb, relb);
        q2b[0] += qcb[0];
        for (int i = 1; i < 4; ++i) q2b[i] -= qcb[i];
        for (int i = 0; i < 3; ++i) { p1tot[i] += relb[i]; x1b[i] += relb[i]; x2b[i] -= relb[i]; }
    }
    double c1b[3] = {0, 0, 0}, c0b[3] = {0, 0, 0};
    rev_quat_apply(q1, c1, p1tot, q1b, c1b);                     // p1 = R1 c1
    rev_query_sdf(S1, c1, 0.0, n1b, c1b, sb1);                   // (d1, n1) at the corrected point; d1 itself is not an output
    double dAb = 0.0, nAb[3];
    for (int i = 0; i < 3; ++i) { c0b[i] += c1b[i]; dAb -= c1b[i] * nA[i]; nAb[i] = -dA * c1b[i]; }   // c1 = c0 - dA nA
    rev_query_sdf(S1, c0, dAb, nAb, c0b, sb1);
    // c0 = sum_v tri_v abc_v, tri_v[i] depends on parameter sel(i) of body 1 with derivative tg[v][i]
    for (int v = 0; v < 3; ++v)
        for (int i = 0; i < 3; ++i) {
            const int s = ty1 == SHAPE_BOX ? i : ((ty1 == SHAPE_CYLINDER && i == 2) ? 1 : 0);
            pr1b[s] += tg[v][i] * abc[v] * c0b[i];
        }
    rev_make_shape(ty1, prm1, S1, sb1, pr1b);
    rev_make_shape(ty2, prm2, S2, sb2, pr2b);
    for (int i = 0; i < 4; ++i) { out[i] = q1b[i]; out[7 + i] = q2b[i]; }
    for (int i = 0; i < 3; ++i) { out[4 + i] = x1b[i]; out[11 + i] = x2b[i]; out[14 + i] = pr1b[i]; out[17 + i] = pr2b[i]; }
}

}  // inline namespace shapes_lean
}  // namespace dss
