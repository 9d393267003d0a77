// contacts2d.hip -- analytic 2-D contacts between circles and convex polygons: the contact handler of the reference's
// 2-D world (DiffContactHandler, lcp_physics/physics/contacts.py:55-357; BASELINE configs[0]), for a batch of body pairs.
//
// One lane per pair.  The contact set is piecewise smooth in the bodies' coordinates: which feature pair touches is decided
// on VALUES (the reference decides on `.item()`), the coordinates of the contact are smooth in between.  The same templated
// code therefore serves the forward (T = double) and the vector-Jacobian product (T = Dual<N>, seeds over the pair's
// coordinates in chunks of N; the branch decisions repeat those of the value pass because they only read values).
//
// Circle against polygon: the reference walks GJK from a random start vertex to the polygon's closest feature
// (contacts.py:85-112).  The closest point of a convex polygon is unique, so the walk's result does not depend on its
// path; it is found here edge by edge, and written with the reference's formula for a point on a segment
// (u s0 + v s1 with the normalised barycentrics of get_barycentric_coords, :335-341) so that the derivative is the
// derivative of the same expression.  A centre inside the polygon takes the separating-axis branch (:121-141).
// Polygon against polygon: separating axes from each side (test_separations, :229-258), reference / incident edge and the
// two clips (:150-206, :260-296).  The cyclic start of the axis loops (`last_sat_idx`) is state of the bodies: in and out.
#include <math.h>

#include "../../include/diffsdfsim_hip.h"
#include "geom.h"

namespace {
using namespace dss;

constexpr int MAXV = DSS_C2D_MAXV;

template <class T> struct Body2 {
    int kind, nv;        // 0 circle, 1 convex polygon (clockwise in the reference's y-down frame, about its centroid)
    T pos[2], rad;
    T v[MAXV][2];
};
template <class T> struct Contact2 { T n[2], p1[2], p2[2], pen; };

template <class T> __host__ __device__ inline T dot2(const T *a, const T *b) { return a[0] * b[0] + a[1] * b[1]; }
template <class T> __host__ __device__ inline T norm2(const T *a) { return t_sqrt(a[0] * a[0] + a[1] * a[1]); }

// outward unit normal of edge i and the edge's length (left_orthogonal(edge) / |edge|, utils.py:124-127)
template <class T> __host__ __device__ inline void edge_normal(const Body2<T> &h, int i, T *n, T &len)
{
    const int j = (i + 1) % h.nv;
    const T e[2] = {h.v[j][0] - h.v[i][0], h.v[j][1] - h.v[i][1]};
    len = norm2(e);
    n[0] = e[1] / len;
    n[1] = -(e[0] / len);
}

template <class T> __host__ __device__ inline int circle_circle(const Body2<T> &a, const Body2<T> &b, double eps, Contact2<T> *out)
{
    const T r = a.rad + b.rad, d[2] = {a.pos[0] - b.pos[0], a.pos[1] - b.pos[1]};
    const T dist = norm2(d), pen = r - dist;
    if (val(pen) < -eps) return 0;
    Contact2<T> &c = out[0];
    for (int i = 0; i < 2; ++i) {
        c.n[i] = d[i] / dist;
        c.p1[i] = -(c.n[i] * (a.rad - pen / 2.0));
        c.p2[i] = c.n[i] * (b.rad - pen / 2.0);
    }
    c.pen = pen;
    return 1;
}

// circle c against polygon h; `swapped`: the circle is the pair's second body (contacts.py:143-146)
template <class T>
__host__ __device__ inline int circle_polygon(const Body2<T> &c, const Body2<T> &h, bool swapped, double eps, int &sat_h, Contact2<T> *out)
{
    const T ctr[2] = {c.pos[0] - h.pos[0], c.pos[1] - h.pos[1]};      // circle centre in the polygon's frame
    bool inside = true;
    for (int i = 0; i < h.nv; ++i) {
        const int j = (i + 1) % h.nv;
        // strictly inside every edge <=> strictly inside some triangle of vertices, the walk's three-point ending
        const double ex = val(h.v[j][0]) - val(h.v[i][0]), ey = val(h.v[j][1]) - val(h.v[i][1]);
        if (ey * (val(ctr[0]) - val(h.v[i][0])) - ex * (val(ctr[1]) - val(h.v[i][1])) >= 0.0) inside = false;
    }
    T nrm[2], pt1[2], pt2[2], dist;
    if (!inside) {
        // closest feature: the nearest of the per-edge closest points (a vertex is reached from either of its edges)
        int be = 0, bk = 0;      // bk: 0 interior of edge be, 1 its first vertex, 2 its second
        double bd = INFINITY;
        for (int i = 0; i < h.nv; ++i) {
            const int j = (i + 1) % h.nv;
            const double ax = val(h.v[i][0]), ay = val(h.v[i][1]), ex = val(h.v[j][0]) - ax, ey = val(h.v[j][1]) - ay;
            const double px = val(ctr[0]) - ax, py = val(ctr[1]) - ay, t = (px * ex + py * ey) / (ex * ex + ey * ey);
            const int k = t <= 0.0 ? 1 : (t >= 1.0 ? 2 : 0);
            const double qx = k == 1 ? 0.0 : (k == 2 ? ex : t * ex), qy = k == 1 ? 0.0 : (k == 2 ? ey : t * ey);
            const double d2 = (px - qx) * (px - qx) + (py - qy) * (py - qy);
            if (d2 < bd) { bd = d2; be = i; bk = k; }
        }
        const int i0 = be, i1 = (be + 1) % h.nv;
        T cl[2];
        if (bk == 0) {
            const T df[2] = {h.v[i1][0] - h.v[i0][0], h.v[i1][1] - h.v[i0][1]};
            const T dn = norm2(df), nd[2] = {df[0] / dn, df[1] / dn};
            const T a[2] = {h.v[i1][0] - ctr[0], h.v[i1][1] - ctr[1]}, b[2] = {ctr[0] - h.v[i0][0], ctr[1] - h.v[i0][1]};
            const T u = dot2(a, nd) / dn, w = dot2(b, nd) / dn;
            for (int k = 0; k < 2; ++k) cl[k] = u * h.v[i0][k] + w * h.v[i1][k];
        } else {
            const int iv = bk == 1 ? i0 : i1;
            for (int k = 0; k < 2; ++k) cl[k] = h.v[iv][k];
        }
        for (int k = 0; k < 2; ++k) { pt2[k] = cl[k]; pt1[k] = (cl[k] + h.pos[k]) - c.pos[k]; }
        const T len = norm2(pt1);
        dist = len - c.rad;
        if (val(dist) > eps) return 0;
        for (int k = 0; k < 2; ++k) nrm[k] = -(pt1[k] / len);       // from the closest point towards the centre
    } else {
        double best = -1e10;
        bool have = false;
        for (int s = 0, start = sat_h; s < h.nv; ++s) {
            const int i = (start + s) % h.nv;
            T n[2], len;
            edge_normal(h, i, n, len);
            const T rel[2] = {ctr[0] - h.v[i][0], ctr[1] - h.v[i][1]};
            const T d = dot2(n, rel) - c.rad;
            if (val(d) > best) {
                sat_h = i;
                if (val(d) > eps) return 0;
                best = val(d); have = true;
                dist = d;
                for (int k = 0; k < 2; ++k) {
                    nrm[k] = n[k];
                    pt2[k] = ctr[k] + n[k] * (-(d + c.rad));
                    pt1[k] = (pt2[k] + h.pos[k]) - c.pos[k];
                }
            }
        }
        if (!have) return 0;
    }
    Contact2<T> &o = out[0];
    for (int k = 0; k < 2; ++k) {
        o.n[k] = swapped ? -nrm[k] : nrm[k];
        o.p1[k] = swapped ? pt2[k] : pt1[k];
        o.p2[k] = swapped ? pt1[k] : pt2[k];
    }
    o.pen = -dist;
    return 1;
}

template <class T> struct Sep { T dist, n[2], len; int vertex, edge; bool apart; };

// the axis of h1 along which h2 is least deep (contacts.py:229-258); n = outward normal of that edge of h1
template <class T> __host__ __device__ inline void separation(const Body2<T> &h1, const Body2<T> &h2, double eps, int start, Sep<T> &r)
{
    double best = -1e10;
    r.apart = false; r.vertex = -1; r.edge = start;
    for (int s = 0; s < h1.nv; ++s) {
        const int i = (start + s) % h1.nv;
        T n[2], len;
        edge_normal(h1, i, n, len);
        // support vertex of h2 against the normal: the LAST of the largest, none below -1 (get_support, :218-227)
        int sv = -1;
        double sb = -1.0;
        for (int k = 0; k < h2.nv; ++k) {
            const double d = -(val(h2.v[k][0]) * val(n[0]) + val(h2.v[k][1]) * val(n[1]));
            if (d >= sb) { sb = d; sv = k; }
        }
        if (sv < 0) { r.apart = true; r.edge = i; return; }
        const T sp[2] = {(h2.v[sv][0] + h2.pos[0]) - h1.pos[0] - h1.v[i][0], (h2.v[sv][1] + h2.pos[1]) - h1.pos[1] - h1.v[i][1]};
        const T d = dot2(n, sp);
        if (val(d) > best) {
            if (val(d) > eps) { r.apart = true; r.edge = i; r.dist = d; return; }
            best = val(d);
            r.dist = d; r.n[0] = n[0]; r.n[1] = n[1]; r.len = len; r.vertex = sv; r.edge = i;
        }
    }
}

// the part of segment (a, b) on the non-negative side of  n . x + off  (clip_segment_to_line, :276-296); count returned
template <class T> __host__ __device__ inline int clip2(const T in[2][2], const T *n, const T &off, T out[2][2])
{
    const T d0 = dot2(n, in[0]) + off, d1 = dot2(n, in[1]) + off;
    int m = 0;
    if (val(d0) >= 0.0) { out[m][0] = in[0][0]; out[m][1] = in[0][1]; ++m; }
    if (val(d1) >= 0.0) { out[m][0] = in[1][0]; out[m][1] = in[1][1]; ++m; }
    if (val(d0) * val(d1) < 0.0 || m < 2) {
        const T t = d0 / (d0 - d1);
        if (m < 2) { out[m][0] = in[0][0] + t * (in[1][0] - in[0][0]); out[m][1] = in[0][1] + t * (in[1][1] - in[0][1]); }
        ++m;
    }
    return m;
}

template <class T>
__host__ __device__ inline int polygon_polygon(const Body2<T> &b1, const Body2<T> &b2, double eps, int &sat1, int &sat2, Contact2<T> *out)
{
    Sep<T> s1, s2;
    separation(b1, b2, eps, sat1, s1);
    sat1 = s1.edge;
    if (s1.apart) return 0;
    separation(b2, b1, eps, sat2, s2);
    sat2 = s2.edge;
    if (s2.apart) return 0;
    const bool ref2 = val(s2.dist) > val(s1.dist);       // the polygon whose edge is the reference face
    const Body2<T> &R = ref2 ? b2 : b1, &I = ref2 ? b1 : b2;
    const Sep<T> &S = ref2 ? s2 : s1;
    const T n[2] = {S.n[0], S.n[1]}, half = S.len / 2.0;
    // incident edge: of the two edges at the support vertex, the one whose normal opposes n most (first on a tie)
    int ie = -1;
    double md = 1e10;
    for (int q = 0; q < 2; ++q) {
        const int i = q == 0 ? (S.vertex + I.nv - 1) % I.nv : S.vertex;
        T m[2], len;
        edge_normal(I, i, m, len);
        const double d = val(n[0]) * val(m[0]) + val(n[1]) * val(m[1]);
        if (d < md) { md = d; ie = i; }
    }
    const int je = (ie + 1) % I.nv;
    T seg[2][2], c1[2][2], c2[2][2];
    for (int k = 0; k < 2; ++k) { seg[0][k] = (I.v[ie][k] + I.pos[k]) - R.pos[k]; seg[1][k] = (I.v[je][k] + I.pos[k]) - R.pos[k]; }
    const T cp[2] = {n[1], -n[0]}, cm[2] = {-n[1], -(-n[0])};
    if (clip2(seg, cp, half, c1) < 2) return 0;
    const int m = clip2(c1, cm, half, c2);
    int cnt = 0;
    for (int q = 0; q < m && q < 2; ++q) {
        const T rel[2] = {c2[q][0] - R.v[S.edge][0], c2[q][1] - R.v[S.edge][1]};
        const T d = dot2(n, rel);
        if (val(d) <= eps) {
            Contact2<T> &o = out[cnt++];
            T on_ref[2], on_inc[2];
            for (int k = 0; k < 2; ++k) { on_ref[k] = c2[q][k] + n[k] * (-d); on_inc[k] = (on_ref[k] + R.pos[k]) - I.pos[k]; }
            for (int k = 0; k < 2; ++k) {
                // (normal, p1, p2): offsets from body 1 and body 2; the normal points from body 2 to body 1
                o.n[k] = ref2 ? n[k] : -n[k];
                o.p1[k] = ref2 ? on_inc[k] : on_ref[k];
                o.p2[k] = ref2 ? on_ref[k] : on_inc[k];
            }
            o.pen = -d;
        }
    }
    return cnt;
}

template <class T>
__host__ __device__ inline int pair_contacts(const Body2<T> &b1, const Body2<T> &b2, double eps, int &sat1, int &sat2, Contact2<T> *out)
{
    if (b1.kind == 0 && b2.kind == 0) return circle_circle(b1, b2, eps, out);
    if (b1.kind == 0) return circle_polygon(b1, b2, false, eps, sat2, out);
    if (b2.kind == 0) return circle_polygon(b2, b1, true, eps, sat1, out);
    return polygon_polygon(b1, b2, eps, sat1, sat2, out);
}

struct Args2D {
    int P, maxv;
    const int *kind, *nv, *sat_in;
    const double *pos, *rad, *verts;
    double eps;
};

// coordinate k of the pair: body s = k / NC, then pos(2), rad, verts(2 maxv)
template <class T> __host__ __device__ inline void load_body(const Args2D &A, int p, int s, Body2<T> &b)
{
    const size_t o = (size_t)s * A.P + p;
    b.kind = A.kind[o]; b.nv = A.nv[o];
    // the C ABI hands these in from device memory: a vertex count beyond the table (MAXV, maxv) would index past v[][] in every
    // loop over the polygon -- clamp it (the forward kernel reports such a pair with count = -1, as it does an unknown kind)
    if (b.nv > MAXV) b.nv = MAXV;
    if (b.nv > A.maxv) b.nv = A.maxv;
    if (b.nv < 0) b.nv = 0;
    b.pos[0] = T(A.pos[2 * o]); b.pos[1] = T(A.pos[2 * o + 1]);
    b.rad = T(A.rad[o]);
    for (int i = 0; i < MAXV; ++i)
        for (int k = 0; k < 2; ++k) b.v[i][k] = T(i < b.nv && i < A.maxv ? A.verts[(o * A.maxv + i) * 2 + k] : 0.0);
}

__global__ void __launch_bounds__(64) contacts2d_forward_kernel(Args2D A, int *sat_out, int *count, double *out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.P) return;
    Body2<double> b1, b2;
    load_body(A, p, 0, b1);
    load_body(A, p, 1, b2);
    int s1 = A.sat_in[p], s2 = A.sat_in[A.P + p];
    Contact2<double> c[2];
    auto bad = [&](int s) {
        const size_t o = (size_t)s * A.P + p;
        const int k = A.kind[o], nv = A.nv[o];
        return (k != 0 && k != 1) || nv < 0 || nv > MAXV || nv > A.maxv;
    };
    if (bad(0) || bad(1)) {      // malformed pair: flagged, no contact written
        count[p] = -1;
        sat_out[p] = s1; sat_out[A.P + p] = s2;
        for (int q = 0; q < 14; ++q) out[(size_t)p * 14 + q] = 0.0;
        return;
    }
    const int n = pair_contacts(b1, b2, A.eps, s1, s2, c);
    count[p] = n;
    sat_out[p] = s1; sat_out[A.P + p] = s2;
    for (int q = 0; q < 2; ++q) {
        double *o = out + ((size_t)p * 2 + q) * 7;
        const bool on = q < n;
        o[0] = on ? c[q].n[0] : 0.0; o[1] = on ? c[q].n[1] : 0.0;
        o[2] = on ? c[q].p1[0] : 0.0; o[3] = on ? c[q].p1[1] : 0.0;
        o[4] = on ? c[q].p2[0] : 0.0; o[5] = on ? c[q].p2[1] : 0.0;
        o[6] = on ? c[q].pen : 0.0;
    }
}

constexpr int NS = 4;      // seeds per pass of the vector-Jacobian product

__host__ __device__ inline void seed_coord(Body2<Dual<NS>> &b1, Body2<Dual<NS>> &b2, int maxv, int k, int slot)
{
    const int NC = 3 + 2 * maxv;
    Body2<Dual<NS>> &b = k < NC ? b1 : b2;
    const int j = k % NC;
    if (j < 2) b.pos[j].d[slot] = 1.0;
    else if (j == 2) b.rad.d[slot] = 1.0;
    else b.v[(j - 3) / 2][(j - 3) % 2].d[slot] = 1.0;
}

__global__ void __launch_bounds__(64)
contacts2d_backward_kernel(Args2D A, const double *gout, double *g_pos, double *g_rad, double *g_verts)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.P) return;
    const int NC = 3 + 2 * A.maxv;
    for (int k0 = 0; k0 < 2 * NC; k0 += NS) {
        Body2<Dual<NS>> b1, b2;
        load_body(A, p, 0, b1);
        load_body(A, p, 1, b2);
        for (int s = 0; s < NS; ++s) if (k0 + s < 2 * NC) seed_coord(b1, b2, A.maxv, k0 + s, s);
        int s1 = A.sat_in[p], s2 = A.sat_in[A.P + p];
        Contact2<Dual<NS>> c[2];
        const int n = pair_contacts(b1, b2, A.eps, s1, s2, c);
        for (int s = 0; s < NS; ++s) {
            const int k = k0 + s;
            if (k >= 2 * NC) break;
            double acc = 0.0;
            for (int q = 0; q < n; ++q) {
                const double *g = gout + ((size_t)p * 2 + q) * 7;
                acc += g[0] * c[q].n[0].d[s] + g[1] * c[q].n[1].d[s] + g[2] * c[q].p1[0].d[s] + g[3] * c[q].p1[1].d[s] +
                       g[4] * c[q].p2[0].d[s] + g[5] * c[q].p2[1].d[s] + g[6] * c[q].pen.d[s];
            }
            const int body = k / NC, j = k % NC;
            const size_t o = (size_t)body * A.P + p;
            if (j < 2) g_pos[2 * o + j] = acc;
            else if (j == 2) g_rad[o] = acc;
            else g_verts[o * A.maxv * 2 + (j - 3)] = acc;
        }
    }
}

int check(int npairs, int maxv)
{
    if (npairs < 0 || maxv < 1) return DSS_E_BADARG;
    if (maxv > MAXV) return DSS_E_UNSUPPORTED;
    return DSS_OK;
}
}  // namespace

extern "C" int dss_contacts2d_forward(int npairs, int maxv, const int *kind, const int *nv, const double *pos, const double *rad,
                                      const double *verts, const int *sat_in, double eps, int *sat_out, int *count, double *out,
                                      void *stream)
{
    if (int rc = check(npairs, maxv)) return rc;
    if (npairs == 0) return DSS_OK;
    const Args2D A{npairs, maxv, kind, nv, sat_in, pos, rad, verts, eps};
    hipLaunchKernelGGL(contacts2d_forward_kernel, dim3((npairs + 63) / 64), dim3(64), 0, (hipStream_t)stream, A, sat_out, count, out);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_BADARG;
}

extern "C" int dss_contacts2d_backward(int npairs, int maxv, const int *kind, const int *nv, const double *pos, const double *rad,
                                       const double *verts, const int *sat_in, double eps, const double *gout, double *g_pos,
                                       double *g_rad, double *g_verts, void *stream)
{
    if (int rc = check(npairs, maxv)) return rc;
    if (npairs == 0) return DSS_OK;
    const Args2D A{npairs, maxv, kind, nv, sat_in, pos, rad, verts, eps};
    hipLaunchKernelGGL(contacts2d_backward_kernel, dim3((npairs + 63) / 64), dim3(64), 0, (hipStream_t)stream, A, gout, g_pos, g_rad, g_verts);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_BADARG;
}
