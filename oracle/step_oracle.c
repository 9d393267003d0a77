/*
 * step_oracle.c -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * Plain-C restatement of ONE SCENE of the reference's whole time step, written from the reference's Python and
 * independently of the HIP kernels (it shares no source with diffsdfsim_amd/csrc):
 *
 *   World.step / step_dt          lcp_physics/physics/world.py:119-139, 241-379   (retry loop, dt halving, TOC dt, escape)
 *   PdipmEngine.solve_dynamics    lcp_physics/physics/engines.py:31-83            (u = M v + dt f, G / F / h assembly)
 *   World3D.M / Jc / Jf           sdf_physics/physics3d/world.py:48-101
 *   World.mu / E / restitutions   lcp_physics/physics/world.py:402-501
 *   Body3D.move / set_p           sdf_physics/physics3d/bodies.py:488-511
 *   SDF3D.query_sdfs, box / sphere / cylinder SDF + grad   sdf_physics/physics3d/bodies.py:38-170, 721-760
 *   FWContactHandler: _overlap, _frank_wolfe (float32 step sizes), _compute_contacts, _filter_contacts, __call__,
 *   _search_contacts              sdf_physics/physics3d/contacts.py:27-272
 *   pytorch3d 0.7.5 transforms (so3_exponential_map, quaternion_*; restated from their documented semantics)
 *   orthogonal()                  sdf_physics/physics3d/utils.py:247-256
 * The LCP itself is oracle/lcp_oracle.c (the reference's dense PDIPM), called with the dense G, F, h of engines.py.
 *
 * Broad phase: all pairs (i < j) in body order, honouring no_contact -- the canonical order of SURVEY.md 8a-R3 (the
 * goldens were recorded with the same stand-in for py3ode's HashSpace).
 * Convex hull of a contact cluster (contacts.py:126-152, scipy's Qhull in the reference): by default an own small hull
 * (`own_hull`), or -- so_set_hull_callback -- a callback that the Python front end points at scipy.spatial.ConvexHull
 * itself, which is what the reference calls; the checker uses the callback, the timing leg the built-in.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.  Pinned against the rollout
 * goldens (tests/test_oracle_step.py), which were produced by the imported reference.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

int lcp_oracle_forward1(const double *Q, const double *p, const double *G, const double *h, const double *A, const double *b,
                        const double *F, int nz, int nineq, int neq, double eps, int not_improved_lim, int max_iter,
                        int check_spd, double *zhat, double *lam, double *slack, double *nu, int *iters, double *best_resid);
int lcp_oracle_backward1(const double *Q, const double *G, const double *A, const double *F, int nz, int nineq, int neq,
                         const double *zhat, const double *lam, const double *slack, const double *nu, const double *dl_dz,
                         double *dQ, double *dp, double *dG, double *dh, double *dA, double *db, double *dF);

enum { SH_BOX = 0, SH_SPHERE = 1, SH_CYL = 2 };
#define LAP_EPS 1e-3 /* _compute_contacts is called without eps: Defaults3D.EPSILON (contacts.py:161, 254-264) */

typedef struct {
    double n[3], p1[3], p2[3], pen;
    int b1, b2;
    int stable;            /* stable_mask of contacts.py:198 (1: body 2's normal) */
    double lap[2];
} contact_t;

typedef struct { contact_t *c; int n, cap; } clist_t;

typedef struct {
    int shape, fixed;
    double prm[3], scale;
    double p[7], v[6];     /* pose: quaternion wxyz + position; velocity: angular, linear */
    double mass, I[9], Mrot[9], rest, fric, fext[6];
    double *verts; int nv; int *faces; int nf; int borrowed;
} body_t;

typedef int (*hull_cb_t)(const double *pts, int m, int dim, int *out);

typedef struct {
    int nb;
    body_t *b;
    unsigned char *nocon;  /* [nb][nb] */
    double dt, eps, tol, t;
    int fric_dirs, strict, toc_diff, max_iter, lcp_bwd;
    clist_t contacts;
    int have_toc; double last_dt;      /* toc_contacts non-empty / last_dt (world.py:253-257, 273-341) */
    /* trajectory: one record per accepted sub-step (world.py:373-377) */
    int nsub, sub_cap; double *tr_t, *tr_p, *tr_v; int *tr_nc; clist_t *tr_c;
    long n_attempts, n_lcp, n_lcp_rows, n_fw_cand;
    double t_solve, t_detect;          /* seconds in solve_dynamics / find_contacts */
    hull_cb_t hull_cb;
    int err;
} world_t;

/* ---- small vector / quaternion helpers (pytorch3d.transforms semantics, real-first quaternions) -------------------- */
static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(const double *a, const double *b, double *o)
{
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static double norm3(const double *a) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
/* torch.nn.functional.normalize(x, dim, eps=1e-12): x / max(|x|, eps) */
static void normalize3(const double *a, double *o)
{
    double n = norm3(a); if (n < 1e-12) n = 1e-12;
    o[0] = a[0] / n; o[1] = a[1] / n; o[2] = a[2] / n;
}
static void qraw_mul(const double *a, const double *b, double *o)
{
    double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
static void qmul(const double *a, const double *b, double *o)      /* quaternion_multiply: standardised (w >= 0) */
{
    qraw_mul(a, b, o);
    if (o[0] < 0) { o[0] = -o[0]; o[1] = -o[1]; o[2] = -o[2]; o[3] = -o[3]; }
}
static void qinv(const double *q, double *o) { o[0] = q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = -q[3]; }
static void qapply(const double *q, const double *p, double *o)    /* quaternion_apply: (q (0,p)) q^-1 */
{
    double pq[4] = {0.0, p[0], p[1], p[2]}, t[4], qi[4], r[4];
    qraw_mul(q, pq, t); qinv(q, qi); qraw_mul(t, qi, r);
    o[0] = r[1]; o[1] = r[2]; o[2] = r[3];
}
static void q2mat(const double *q, double *m)
{
    double r = q[0], i = q[1], j = q[2], k = q[3], s = 2.0 / (r * r + i * i + j * j + k * k);
    m[0] = 1 - s * (j * j + k * k); m[1] = s * (i * j - k * r); m[2] = s * (i * k + j * r);
    m[3] = s * (i * j + k * r); m[4] = 1 - s * (i * i + k * k); m[5] = s * (j * k - i * r);
    m[6] = s * (i * k - j * r); m[7] = s * (j * k + i * r); m[8] = 1 - s * (i * i + j * j);
}
static void so3_exp(const double *w, double *R)                     /* so3_exponential_map(v, eps = 1e-4) */
{
    double nr = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double ang = sqrt(nr < 1e-4 ? 1e-4 : nr), inv = 1.0 / ang, f1 = inv * sin(ang), f2 = inv * inv * (1.0 - cos(ang));
    double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}, K2[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += K[3 * i + k] * K[3 * k + j]; K2[3 * i + j] = s; }
    for (int i = 0; i < 9; ++i) R[i] = f1 * K[i] + f2 * K2[i] + ((i % 4 == 0) ? 1.0 : 0.0);
}
static void mat2q(const double *m, double *q)                       /* matrix_to_quaternion */
{
    double m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6], m21 = m[7], m22 = m[8];
    double a[4] = {1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22}, qa[4];
    for (int i = 0; i < 4; ++i) qa[i] = a[i] > 0 ? sqrt(a[i]) : 0.0;
    double c[4][4] = {{qa[0] * qa[0], m21 - m12, m02 - m20, m10 - m01}, {m21 - m12, qa[1] * qa[1], m10 + m01, m02 + m20},
                      {m02 - m20, m10 + m01, qa[2] * qa[2], m12 + m21}, {m10 - m01, m20 + m02, m21 + m12, qa[3] * qa[3]}};
    int best = 0; for (int i = 1; i < 4; ++i) if (qa[i] > qa[best]) best = i;
    double den = 2.0 * (qa[best] > 0.1 ? qa[best] : 0.1);
    for (int i = 0; i < 4; ++i) q[i] = c[best][i] / den;
    if (q[0] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
}

/* ---- SDF queries (bodies.py:38-170, 721-760) -------------------------------------------------------------------------- */
static double fmax2(double a, double b) { return a > b ? a : b; }

static double prim_sdf(const body_t *B, const double *x)           /* sdf_func(pts / scale, params / scale) */
{
    if (B->shape == SH_BOX) {
        double q[3], md, m2 = 0;
        for (int i = 0; i < 3; ++i) q[i] = fabs(x[i]) - (B->prm[i] / B->scale) / 2;
        md = fmax2(fmax2(q[0], q[1]), q[2]);
        for (int i = 0; i < 3; ++i) { double m = q[i] > 0 ? q[i] : 0.0; m2 += m * m; }
        return sqrt(m2) + (md < 0 ? md : 0.0);
    } else if (B->shape == SH_SPHERE) {
        return norm3(x) - B->prm[0] / B->scale;
    } else {
        double ps[2] = {sqrt(x[0] * x[0] + x[1] * x[1]), x[2]};
        double q[2] = {fabs(ps[0]) - B->prm[0] / B->scale, fabs(ps[1]) - (B->prm[1] / B->scale) / 2};
        double md = fmax2(q[0], q[1]), m0 = q[0] > 0 ? q[0] : 0, m1 = q[1] > 0 ? q[1] : 0;
        return sqrt(m0 * m0 + m1 * m1) + (md < 0 ? md : 0.0);
    }
}
static void prim_grad(const body_t *B, const double *x, double *g)  /* grad_func(pts / scale, params / scale), un-normalised */
{
    if (B->shape == SH_BOX) {                                        /* bodies.py:52-72 */
        double q[3], sg[3], md, m[3], mn[3], go[3];
        for (int i = 0; i < 3; ++i) { q[i] = fabs(x[i]) - (B->prm[i] / B->scale) / 2; sg[i] = x[i] > 0 ? 1.0 : (x[i] < 0 ? -1.0 : 1.0); }
        md = fmax2(fmax2(q[0], q[1]), q[2]);
        for (int i = 0; i < 3; ++i) m[i] = q[i] > 0 ? q[i] : 0.0;
        normalize3(m, mn);
        for (int i = 0; i < 3; ++i) go[i] = (mn[i] + (md <= 0 ? 1.0 : 0.0) * (q[i] == md ? 1.0 : 0.0)) * sg[i];
        normalize3(go, g);
    } else if (B->shape == SH_SPHERE) {
        normalize3(x, g);
    } else {                                                         /* cylinder_sdf_grad, bodies.py:104-129 */
        double rho = sqrt(x[0] * x[0] + x[1] * x[1]);
        double q[2] = {fabs(rho) - B->prm[0] / B->scale, fabs(x[2]) - (B->prm[1] / B->scale) / 2};
        double md = fmax2(q[0], q[1]), m[2] = {q[0] > 0 ? q[0] : 0, q[1] > 0 ? q[1] : 0};
        double mn = sqrt(m[0] * m[0] + m[1] * m[1]); if (mn < 1e-12) mn = 1e-12;
        double g2[2] = {m[0] / mn + (md <= 0 ? 1.0 : 0.0) * (q[0] == md ? 1.0 : 0.0), m[1] / mn + (md <= 0 ? 1.0 : 0.0) * (q[1] == md ? 1.0 : 0.0)};
        double xy[3] = {x[0], x[1], 0.0}, d[3];
        normalize3(xy, d);
        double sz = x[2] > 0 ? 1.0 : (x[2] < 0 ? -1.0 : 1.0);
        double go[3] = {g2[0] * d[0], g2[0] * d[1], g2[1] * sz};
        normalize3(go, g);
    }
}
/* SDF3D.query_sdfs: outside the cube |x| <= scale the value is `scale` and the gradient zero */
static double query(const body_t *B, const double *x, double *grad)
{
    int in = fabs(x[0]) <= B->scale && fabs(x[1]) <= B->scale && fabs(x[2]) <= B->scale;
    if (grad) grad[0] = grad[1] = grad[2] = 0.0;
    if (!in) return 1.0 * B->scale;
    double xi[3] = {x[0] / B->scale, x[1] / B->scale, x[2] / B->scale};
    if (grad) { double g[3]; prim_grad(B, xi, g); normalize3(g, grad); }
    return prim_sdf(B, xi) * B->scale;
}

/* ---- contact lists ---------------------------------------------------------------------------------------------------- */
static void cl_push(clist_t *L, const contact_t *c)
{
    if (L->n == L->cap) { L->cap = L->cap ? 2 * L->cap : 64; L->c = realloc(L->c, sizeof(contact_t) * L->cap); }
    L->c[L->n++] = *c;
}
static void cl_copy(clist_t *d, const clist_t *s)
{
    d->n = 0;
    for (int i = 0; i < s->n; ++i) cl_push(d, &s->c[i]);
}

/* ---- hull of a contact cluster (contacts.py:126-152) ------------------------------------------------------------------- */
static double g_cmp_tol;     /* qsort has no context argument; hull2 is only ever entered under its own omp critical section */
static int cmp_xy(const void *a, const void *b)
{
    /* x within the collinearity tolerance counts as equal: on a (nearly) vertical edge the order must be the order along
       the edge, not the order of the rounding noise in x -- the chain below drops the MIDDLE one of three collinear points */
    const double *p = a, *q = b;
    if (fabs(p[0] - q[0]) > g_cmp_tol) return p[0] < q[0] ? -1 : 1;
    if (fabs(p[1] - q[1]) > g_cmp_tol) return p[1] < q[1] ? -1 : 1;
    return p[2] < q[2] ? -1 : (p[2] > q[2]);
}
/* 2-D hull vertices (counter-clockwise), points nearer than ctol to the chord of their neighbours dropped.  -1: degenerate
   (fewer than 3 points, or all within ftol of a line) -- where Qhull raises QhullError. */
static int hull2(const double *pts, int m, double ftol, double ctol, int *out)
{
    if (m < 3) return -1;
    double *s = malloc(sizeof(double) * 3 * m);
    for (int i = 0; i < m; ++i) { s[3 * i] = pts[2 * i]; s[3 * i + 1] = pts[2 * i + 1]; s[3 * i + 2] = i; }
#pragma omp critical(so_hull2_sort)
    { g_cmp_tol = 8 * ctol; qsort(s, m, 3 * sizeof(double), cmp_xy); }
    /* degenerate: every point within ftol of the line through the two lexicographic extremes */
    {
        double dx = s[3 * (m - 1)] - s[0], dy = s[3 * (m - 1) + 1] - s[1], len = sqrt(dx * dx + dy * dy), far = 0;
        if (len > 0) for (int i = 0; i < m; ++i) { double d = fabs((s[3 * i] - s[0]) * dy - (s[3 * i + 1] - s[1]) * dx) / len; if (d > far) far = d; }
        if (len == 0 || far <= ftol) { free(s); return -1; }
    }
    int *st = malloc(sizeof(int) * 2 * m), k = 0;
#define LEFT(a, b, c) ((s[3 * (b)] - s[3 * (a)]) * (s[3 * (c) + 1] - s[3 * (a) + 1]) - (s[3 * (b) + 1] - s[3 * (a) + 1]) * (s[3 * (c)] - s[3 * (a)]))
#define CHORD(a, c) sqrt((s[3 * (c)] - s[3 * (a)]) * (s[3 * (c)] - s[3 * (a)]) + (s[3 * (c) + 1] - s[3 * (a) + 1]) * (s[3 * (c) + 1] - s[3 * (a) + 1]))
    for (int i = 0; i < m; ++i) {          /* lower chain */
        while (k >= 2) { double ch = CHORD(st[k - 2], i); if (ch == 0 || LEFT(st[k - 2], st[k - 1], i) / ch <= ctol) --k; else break; }
        st[k++] = i;
    }
    for (int i = m - 2, t = k + 1; i >= 0; --i) {   /* upper chain */
        while (k >= t) { double ch = CHORD(st[k - 2], i); if (ch == 0 || LEFT(st[k - 2], st[k - 1], i) / ch <= ctol) --k; else break; }
        st[k++] = i;
    }
    --k;                                     /* last point repeats the first */
#undef LEFT
#undef CHORD
    for (int i = 0; i < k; ++i) out[i] = (int)s[3 * st[i] + 2];
    free(st); free(s);
    return k;
}
/* 3-D hull vertices in input order; -1 where Qhull raises (fewer than 4 points, or flat); -2: too big for this hull */
static int hull3(const double *p, int m, double ftol, double ctol, int *out)
{
    if (m < 4) return -1;
    /* flatness as Qhull's initial simplex sees it: two far points, the point farthest from their line, the point farthest
       from the plane of the three */
    int i0 = 0, i1 = 0, i2 = -1;
    for (int i = 1; i < m; ++i) if (p[3 * i] < p[3 * i0]) i0 = i;
    double best = -1;
    for (int i = 0; i < m; ++i) { double d[3] = {p[3 * i] - p[3 * i0], p[3 * i + 1] - p[3 * i0 + 1], p[3 * i + 2] - p[3 * i0 + 2]}; double l = norm3(d); if (l > best) { best = l; i1 = i; } }
    if (best <= ftol) return -1;
    double e[3] = {p[3 * i1] - p[3 * i0], p[3 * i1 + 1] - p[3 * i0 + 1], p[3 * i1 + 2] - p[3 * i0 + 2]}, el = norm3(e), nrm[3];
    best = -1;
    for (int i = 0; i < m; ++i) { double d[3] = {p[3 * i] - p[3 * i0], p[3 * i + 1] - p[3 * i0 + 1], p[3 * i + 2] - p[3 * i0 + 2]}, c[3]; cross3(e, d, c); double l = norm3(c) / el; if (l > best) { best = l; i2 = i; } }
    if (best <= ftol) return -1;
    { double d[3] = {p[3 * i2] - p[3 * i0], p[3 * i2 + 1] - p[3 * i0 + 1], p[3 * i2 + 2] - p[3 * i0 + 2]}; cross3(e, d, nrm); double l = norm3(nrm); for (int k = 0; k < 3; ++k) nrm[k] /= l; }
    best = -1;
    for (int i = 0; i < m; ++i) { double d[3] = {p[3 * i] - p[3 * i0], p[3 * i + 1] - p[3 * i0 + 1], p[3 * i + 2] - p[3 * i0 + 2]}; double l = fabs(dot3(nrm, d)); if (l > best) best = l; }
    if (best <= ftol) return -1;
    if (m > 160) return -2;
    /* brute force: every supporting plane through three points; the vertices of a facet are the vertices of the 2-D hull of
       the points in its plane (points inside a facet or inside an edge are vertices of no facet) */
    unsigned char *isv = calloc(m, 1), *done = calloc((size_t)m * m, 1);
    double *pl = malloc(sizeof(double) * 2 * m); int *idx = malloc(sizeof(int) * m), *ho = malloc(sizeof(int) * m);
    for (int i = 0; i < m; ++i) for (int j = i + 1; j < m; ++j) for (int k = j + 1; k < m; ++k) {
        if (done[(size_t)i * m + j] && done[(size_t)i * m + k] && done[(size_t)j * m + k]) continue;   /* an edge triple of a facet already walked */
        double a[3] = {p[3 * j] - p[3 * i], p[3 * j + 1] - p[3 * i + 1], p[3 * j + 2] - p[3 * i + 2]};
        double b[3] = {p[3 * k] - p[3 * i], p[3 * k + 1] - p[3 * i + 1], p[3 * k + 2] - p[3 * i + 2]}, n[3];
        cross3(a, b, n);
        double nl = norm3(n), al = norm3(a), bl = norm3(b);
        if (al <= ctol || bl <= ctol || nl / (al > bl ? al : bl) <= ctol) continue;       /* (nearly) collinear or coincident */
        for (int q = 0; q < 3; ++q) n[q] /= nl;
        int pos = 0, neg = 0, nin = 0;
        for (int q = 0; q < m && !(pos && neg); ++q) {
            double d[3] = {p[3 * q] - p[3 * i], p[3 * q + 1] - p[3 * i + 1], p[3 * q + 2] - p[3 * i + 2]}, s = dot3(n, d);
            if (s > ctol) pos = 1; else if (s < -ctol) neg = 1; else idx[nin++] = q;
        }
        if (pos && neg) continue;
        /* plane coordinates */
        double u[3], w[3];
        for (int q = 0; q < 3; ++q) u[q] = a[q] / al;
        cross3(n, u, w);
        for (int q = 0; q < nin; ++q) {
            double d[3] = {p[3 * idx[q]] - p[3 * i], p[3 * idx[q] + 1] - p[3 * i + 1], p[3 * idx[q] + 2] - p[3 * i + 2]};
            pl[2 * q] = dot3(u, d); pl[2 * q + 1] = dot3(w, d);
        }
        int nh = hull2(pl, nin, ctol, ctol, ho);
        for (int q = 0; q < nh; ++q) isv[idx[ho[q]]] = 1;
        for (int q = 0; q < nin; ++q) for (int r = q + 1; r < nin; ++r) done[(size_t)idx[q] * m + idx[r]] = 1;
    }
    int nv = 0;
    for (int i = 0; i < m; ++i) if (isv[i]) out[nv++] = i;
    free(isv); free(done); free(pl); free(idx); free(ho);
    return nv;
}
static int own_hull(const double *pts, int m, int dim, int *out)
{
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, ext = 0;
    for (int i = 0; i < m; ++i) for (int k = 0; k < dim; ++k) { double v = pts[dim * i + k]; if (v < lo[k]) lo[k] = v; if (v > hi[k]) hi[k] = v; }
    for (int k = 0; k < dim; ++k) if (hi[k] - lo[k] > ext) ext = hi[k] - lo[k];
    for (int k = 0; k < dim; ++k) { if (fabs(lo[k]) > ext) ext = fabs(lo[k]); if (fabs(hi[k]) > ext) ext = fabs(hi[k]); }
    /* flat / collinear below 1e-11 of the extent (Qhull: a few ulps of the largest coordinate; exactly flat contact patches are
       off their plane by rounding, ~1e-16, curved ones by >= 1e-7); a vertex must clear its neighbours' chord by 1e-14 */
    const double ftol = 1e-11 * ext, ctol = 1e-14 * ext;
    /* coincident points (neighbouring faces of a mesh converge to shared vertices) are one vertex: the first of them */
    int *uid = malloc(sizeof(int) * m), nu = 0;
    double *up = malloc(sizeof(double) * dim * m);
    for (int i = 0; i < m; ++i) {
        int dup = 0;
        for (int j = 0; j < nu && !dup; ++j) {
            double d = 0;
            for (int k = 0; k < dim; ++k) d = fmax2(d, fabs(pts[dim * i + k] - up[dim * j + k]));
            if (d <= ctol) dup = 1;
        }
        if (!dup) { for (int k = 0; k < dim; ++k) up[dim * nu + k] = pts[dim * i + k]; uid[nu++] = i; }
    }
    int n;
    if (m < dim + 1) n = -1;             /* Qhull counts the input points, duplicates included */
    else n = dim == 3 ? hull3(up, nu, ftol, ctol, out) : hull2(up, nu, ftol, ctol, out);
    for (int i = 0; i < n; ++i) out[i] = uid[out[i]];
    free(uid); free(up);
    return n;
}

int so_own_hull(const double *pts, int m, int dim, int *out) { return own_hull(pts, m, dim, out); }

/* _filter_contacts (contacts.py:97-158): indices into the pair's unfiltered contacts, cluster by cluster */
static int filter_contacts(world_t *W, const contact_t *c, int n, int *keep)
{
    if (n <= 1) { for (int i = 0; i < n; ++i) keep[i] = i; return n; }
    int *ids = malloc(sizeof(int) * n), *cl = malloc(sizeof(int) * n), *ho = malloc(sizeof(int) * n), nk = 0, nv = 0;
    double *ps = malloc(sizeof(double) * 3 * n), *pr = malloc(sizeof(double) * 3 * n);
    for (int i = 0; i < n; ++i) if (norm3(c[i].n) > 1e-12) ids[nv++] = i;
    while (nv > 0) {
        const double *n0 = c[ids[0]].n;
        int m = 0, rest = 0;
        for (int i = 0; i < nv; ++i) {
            double d = dot3(c[ids[i]].n, n0);           /* normals @ n */
            if (acos(d < 1.0 ? d : 1.0) < 1e-2) cl[m++] = ids[i]; else ids[rest++] = ids[i];
        }
        nv = rest;
        for (int i = 0; i < m; ++i) for (int k = 0; k < 3; ++k) ps[3 * i + k] = c[cl[i]].p1[k];
        int cols[3] = {0, 1, 2}, dim = 3, nh = -1;
        while (dim > 1) {
            for (int i = 0; i < m; ++i) for (int k = 0; k < dim; ++k) pr[dim * i + k] = ps[3 * i + cols[k]];
            nh = W->hull_cb ? W->hull_cb(pr, m, dim, ho) : own_hull(pr, m, dim, ho);
            if (getenv("SO_DEBUG_HULL")) { fprintf(stderr, "hull dim %d m %d -> %d:", dim, m, nh); for (int i = 0; i < m; ++i) { fprintf(stderr, " ("); for (int k = 0; k < dim; ++k) fprintf(stderr, "%.17g,", pr[dim * i + k]); fprintf(stderr, ")"); } fprintf(stderr, "\n"); }
            if (nh == -2) { W->err = 3; nh = 0; break; }
            if (nh >= 0) break;
            /* QhullError: drop the dimension of smallest (unbiased) variance; with one point every variance is NaN and
               torch's argmin returns the first */
            int drop = 0; double vbest = INFINITY;
            if (m > 1)
                for (int k = 0; k < dim; ++k) {
                    double mean = 0, var = 0;
                    for (int i = 0; i < m; ++i) mean += ps[3 * i + cols[k]];
                    mean /= m;
                    for (int i = 0; i < m; ++i) { double d = ps[3 * i + cols[k]] - mean; var += d * d; }
                    var /= (m - 1);
                    if (var < vbest) { vbest = var; drop = k; }
                }
            for (int k = drop; k + 1 < dim; ++k) cols[k] = cols[k + 1];
            --dim;
        }
        if (nh < 0) {                                    /* 1-D: min and max (contacts.py:143-150) */
            int imin = 0, imax = 0;
            for (int i = 1; i < m; ++i) { double v = ps[3 * i + cols[0]]; if (v < ps[3 * imin + cols[0]]) imin = i; if (v > ps[3 * imax + cols[0]]) imax = i; }
            ho[0] = imin; nh = 1;
            if (ps[3 * imax + cols[0]] - ps[3 * imin + cols[0]] > W->eps) { ho[1] = imax; nh = 2; }
        }
        for (int i = 0; i < nh; ++i) keep[nk++] = cl[ho[i]];
    }
    free(ids); free(cl); free(ho); free(ps); free(pr);
    return nk;
}

/* ---- contact geometry of one (face, barycentrics) (contacts.py:161-214) ------------------------------------------------- */
static void compute_contact(const body_t *b1, const body_t *b2, int face, const double *abc, contact_t *o)
{
    const int *f = b1->faces + 3 * face;
    double cp1[3], d1, n1[3], cpw[3], q2i[4], cp2[3], d2, n2[3], t[3];
    for (int k = 0; k < 3; ++k) cp1[k] = (b1->verts[3 * f[0] + k] * abc[0] + b1->verts[3 * f[1] + k] * abc[1]) + b1->verts[3 * f[2] + k] * abc[2];
    d1 = query(b1, cp1, n1);
    for (int k = 0; k < 3; ++k) cp1[k] = cp1[k] - d1 * n1[k];
    d1 = query(b1, cp1, n1);
    qapply(b1->p, cp1, cpw);
    for (int k = 0; k < 3; ++k) { cpw[k] += b1->p[4 + k]; t[k] = cpw[k] - b2->p[4 + k]; }
    qinv(b2->p, q2i);
    qapply(q2i, t, cp2);
    d2 = query(b2, cp2, n2);
    double lap1 = 0, lap2 = 0;
    for (int i = 0; i < 3; ++i) {
        double a[3] = {cp1[0], cp1[1], cp1[2]}, b[3] = {cp1[0], cp1[1], cp1[2]};
        a[i] += LAP_EPS; b[i] -= LAP_EPS;
        lap1 += (query(b1, a, NULL) - 2 * d1) + query(b1, b, NULL);
    }
    for (int i = 0; i < 3; ++i) {
        double a[3] = {cp2[0], cp2[1], cp2[2]}, b[3] = {cp2[0], cp2[1], cp2[2]};
        a[i] += LAP_EPS; b[i] -= LAP_EPS;
        lap2 += (query(b2, a, NULL) - 2 * d2) + query(b2, b, NULL);
    }
    o->stable = fabs(lap2) < fabs(lap1);
    o->lap[0] = fabs(lap1); o->lap[1] = fabs(lap2);
    double r2n[3], r1n[3];
    qapply(b2->p, n2, r2n); qapply(b1->p, n1, r1n);
    for (int k = 0; k < 3; ++k) o->n[k] = r2n[k] * (o->stable ? 1.0 : 0.0) - r1n[k] * (o->stable ? 0.0 : 1.0);
    for (int k = 0; k < 3; ++k) t[k] = cp2[k] - d2 * n2[k];
    qapply(b2->p, t, o->p2);
    qapply(b1->p, cp1, o->p1);
    o->pen = -d2;
}

/* ---- _frank_wolfe + _search_contacts (contacts.py:39-94, 249-272): mesh of body i1 against the SDF of body i2 ---------- */
static int search_contacts(world_t *W, int i1, int i2)
{
    const body_t *b1 = &W->b[i1], *b2 = &W->b[i2];
    const double eps = W->eps, tol = W->tol;
    double q2i[4];
    qinv(b2->p, q2i);
    /* vertices of b1 in b2's frame */
    double *v2 = malloc(sizeof(double) * 3 * b1->nv);
    for (int i = 0; i < b1->nv; ++i) {
        double w[3], t[3];
        qapply(b1->p, b1->verts + 3 * i, w);
        for (int k = 0; k < 3; ++k) t[k] = (w[k] + b1->p[4 + k]) - b2->p[4 + k];
        qapply(q2i, t, v2 + 3 * i);
    }
    /* candidate faces: centroid SDF below bounding radius + eps, non-zero gradient */
    int ncand = 0, ccap = 256, *cf = malloc(sizeof(int) * ccap);
    for (int f = 0; f < b1->nf; ++f) {
        const int *fi = b1->faces + 3 * f;
        double x[3], g[3], rad = 0;
        for (int k = 0; k < 3; ++k) x[k] = ((v2[3 * fi[0] + k] + v2[3 * fi[1] + k]) + v2[3 * fi[2] + k]) / 3;
        double sd = query(b2, x, g);
        for (int i = 0; i < 3; ++i) { double d[3] = {x[0] - v2[3 * fi[i]], x[1] - v2[3 * fi[i] + 1], x[2] - v2[3 * fi[i] + 2]}; double r = norm3(d); if (r > rad) rad = r; }
        if (sd < rad + eps && norm3(g) > 1e-12) {
            if (ncand == ccap) { ccap *= 2; cf = realloc(cf, sizeof(int) * ccap); }
            cf[ncand++] = f;
        }
    }
    W->n_fw_cand += ncand;
    if (ncand == 0) { free(v2); free(cf); return 1; }       /* no contacts: all(pens <= tol) of an empty set */
    double *x = malloc(sizeof(double) * 3 * ncand), *abc = calloc(3 * (size_t)ncand, sizeof(double));
    float *gam = malloc(sizeof(float) * ncand); int *ind = malloc(sizeof(int) * ncand);
    double *sd = malloc(sizeof(double) * ncand), *gr = malloc(sizeof(double) * 3 * ncand);
    for (int c = 0; c < ncand; ++c) {
        const int *fi = b1->faces + 3 * cf[c];
        int best = 0; double sb = 0;
        for (int i = 0; i < 3; ++i) { double s = query(b2, v2 + 3 * fi[i], NULL); if (i == 0 || s < sb) { sb = s; best = i; } }
        for (int k = 0; k < 3; ++k) x[3 * c + k] = v2[3 * fi[best] + k];
        abc[3 * c + best] = 1.0;
    }
    for (int iter = 0; iter < 32; ++iter) {
        int all_zero = 1, any_pen = 0;
        const double gamma = 2.0 / (iter + 2.0);
        for (int c = 0; c < ncand; ++c) {
            const int *fi = b1->faces + 3 * cf[c];
            sd[c] = query(b2, x + 3 * c, gr + 3 * c);
            int best = 0; double db = 0;
            for (int i = 0; i < 3; ++i) { double d = dot3(v2 + 3 * fi[i], gr + 3 * c); if (i == 0 || d < db) { db = d; best = i; } }
            ind[c] = best;
            const double *s = v2 + 3 * fi[best];
            double xs[3] = {x[3 * c] - s[0], x[3 * c + 1] - s[1], x[3 * c + 2] - s[2]};
            double impr = dot3(xs, gr + 3 * c);
            /* `gamma * (impr.abs() > tol)`: a Python float times a bool tensor is a float32 tensor (contacts.py:72-73) */
            gam[c] = (float)gamma * (fabs(impr) > tol ? 1.0f : 0.0f);
            if (gam[c] != 0.0f) all_zero = 0;
            if (sd[c] < -tol) any_pen = 1;
        }
        if (all_zero || any_pen) break;
        for (int c = 0; c < ncand; ++c) {
            const int *fi = b1->faces + 3 * cf[c];
            const double *s = v2 + 3 * fi[ind[c]];
            const double g = (double)gam[c], omg = (double)(1.0f - gam[c]);   /* (1.0 - gamma) is formed in float32 too */
            for (int k = 0; k < 3; ++k) x[3 * c + k] = omg * x[3 * c + k] + g * s[k];
            for (int k = 0; k < 3; ++k) abc[3 * c + k] *= omg;
            abc[3 * c + ind[c]] += g;
        }
    }
    /* push x from the triangle to b1's surface; keep what is within eps of b2 */
    double q12[4];
    qmul(q2i, b1->p, q12);
    clist_t raw = {0};
    int *rf = malloc(sizeof(int) * ncand); double *rabc = malloc(sizeof(double) * 3 * ncand);
    for (int c = 0; c < ncand; ++c) {
        const int *fi = b1->faces + 3 * cf[c];
        double xb1[3], g1[3], r[3];
        for (int k = 0; k < 3; ++k) xb1[k] = (b1->verts[3 * fi[0] + k] * abc[3 * c] + b1->verts[3 * fi[1] + k] * abc[3 * c + 1]) + b1->verts[3 * fi[2] + k] * abc[3 * c + 2];
        double s1 = query(b1, xb1, g1);
        qapply(q12, g1, r);
        double xx[3] = {x[3 * c] - s1 * r[0], x[3 * c + 1] - s1 * r[1], x[3 * c + 2] - s1 * r[2]};
        if (getenv("SO_DEBUG")) fprintf(stderr, "fw %d->%d face %d abc %.17g %.17g %.17g sdf %.17g\n", i1, i2, cf[c], abc[3 * c], abc[3 * c + 1], abc[3 * c + 2], query(b2, xx, NULL));
        if (query(b2, xx, NULL) <= eps) {
            contact_t ct; memset(&ct, 0, sizeof ct);
            compute_contact(b1, b2, cf[c], abc + 3 * c, &ct);
            ct.b1 = i1; ct.b2 = i2;
            rf[raw.n] = cf[c]; memcpy(rabc + 3 * raw.n, abc + 3 * c, 3 * sizeof(double));
            cl_push(&raw, &ct);
        }
    }
    int valid = 1;
    for (int i = 0; i < raw.n; ++i) if (!(raw.c[i].pen <= tol)) valid = 0;
    if (valid) {
        int *keep = malloc(sizeof(int) * (raw.n + 1));
        int nk = filter_contacts(W, raw.c, raw.n, keep);
        for (int i = 0; i < nk; ++i) cl_push(&W->contacts, &raw.c[keep[i]]);   /* the second _compute_contacts repeats the same values */
        free(keep);
    } else {
        for (int i = 0; i < raw.n; ++i) cl_push(&W->contacts, &raw.c[i]);
    }
    free(raw.c); free(rf); free(rabc); free(v2); free(cf); free(x); free(abc); free(gam); free(ind); free(sd); free(gr);
    return valid;
}

/* _overlap (contacts.py:27-36): a vertex of each inside the other's query cube */
static int any_vertex_in_cube(const body_t *a, const body_t *b)
{
    double qi[4]; qinv(b->p, qi);
    for (int i = 0; i < a->nv; ++i) {
        double w[3], t[3], l[3];
        qapply(a->p, a->verts + 3 * i, w);
        for (int k = 0; k < 3; ++k) t[k] = (w[k] + a->p[4 + k]) - b->p[4 + k];
        qapply(qi, t, l);
        if (-b->scale <= l[0] && l[0] <= b->scale && -b->scale <= l[1] && l[1] <= b->scale && -b->scale <= l[2] && l[2] <= b->scale) return 1;
    }
    return 0;
}
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

static void find_contacts(world_t *W)
{
    const double t0 = now_s();
    W->contacts.n = 0;
    for (int i = 0; i < W->nb; ++i)
        for (int j = i + 1; j < W->nb; ++j) {
            if (W->nocon[j * W->nb + i]) continue;               /* `geom1 in geom2.no_contact` */
            if (!(any_vertex_in_cube(&W->b[i], &W->b[j]) && any_vertex_in_cube(&W->b[j], &W->b[i]))) continue;
            if (search_contacts(W, i, j)) search_contacts(W, j, i);
        }
    W->t_detect += now_s() - t0;
}

/* ---- dynamics ---------------------------------------------------------------------------------------------------------- */
static void set_p(body_t *B, const double *p)                     /* Body3D.set_p: M <- R I R^T */
{
    memcpy(B->p, p, 7 * sizeof(double));
    double R[9], t[9];
    q2mat(B->p, R);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += R[3 * i + k] * B->I[3 * k + j]; t[3 * i + j] = s; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += t[3 * i + k] * R[3 * j + k]; B->Mrot[3 * i + j] = s; }
}
static void move_body(body_t *B, double dt)                       /* Body3D.move (bodies.py:488-496) */
{
    double w[3] = {B->v[0] * dt, B->v[1] * dt, B->v[2] * dt}, R[9], dq[4], np[7];
    so3_exp(w, R); mat2q(R, dq);
    qmul(dq, B->p, np);
    for (int k = 0; k < 3; ++k) np[4 + k] = B->p[4 + k] + B->v[3 + k] * dt;
    set_p(B, np);
}
/* in-place LU solve of a small dense system (torch.inverse(P) @ u of engines.py:40-54) */
static int dense_solve(int n, double *a, double *b)
{
    for (int k = 0; k < n; ++k) {
        int p = k; double best = fabs(a[k * n + k]);
        for (int i = k + 1; i < n; ++i) if (fabs(a[i * n + k]) > best) { best = fabs(a[i * n + k]); p = i; }
        if (best == 0) return 1;
        if (p != k) { for (int j = 0; j < n; ++j) { double t = a[k * n + j]; a[k * n + j] = a[p * n + j]; a[p * n + j] = t; } double t = b[k]; b[k] = b[p]; b[p] = t; }
        for (int i = k + 1; i < n; ++i) { double l = a[i * n + k] / a[k * n + k]; for (int j = k; j < n; ++j) a[i * n + j] -= l * a[k * n + j]; b[i] -= l * b[k]; }
    }
    for (int i = n - 1; i >= 0; --i) { for (int j = i + 1; j < n; ++j) b[i] -= a[i * n + j] * b[j]; b[i] /= a[i * n + i]; }
    return 0;
}
static void orthogonal(const double *v, double *o)                /* utils.py:247-256 */
{
    int k = 0;
    for (int i = 1; i < 3; ++i) if (fabs(v[i]) < fabs(v[k])) k = i;
    double e[3] = {0, 0, 0}; e[k] = 1.0;
    cross3(e, v, o);
}

/* PdipmEngine.solve_dynamics (engines.py:31-83) with World3D.Jc / Jf (physics3d/world.py:56-101) */
static int solve_dynamics(world_t *W, double dt, double *newv)
{
    const int nb = W->nb, nz = 6 * nb, nc = W->contacts.n, fd = W->fric_dirs;
    int neq = 0;
    for (int i = 0; i < nb; ++i) if (W->b[i].fixed) neq += 6;
    double *M = calloc((size_t)nz * nz, sizeof(double)), *u = calloc(nz + neq, sizeof(double)), *Je = calloc((size_t)(neq + 1) * nz, sizeof(double));
    for (int i = 0, r = 0; i < nb; ++i) {
        const body_t *B = &W->b[i];
        for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) M[(6 * i + a) * nz + 6 * i + c] = B->Mrot[3 * a + c];
        for (int a = 0; a < 3; ++a) M[(6 * i + 3 + a) * nz + 6 * i + 3 + a] = B->mass;
        if (B->fixed) { for (int a = 0; a < 6; ++a) Je[(r + a) * nz + 6 * i + a] = 1.0; r += 6; }
    }
    for (int i = 0; i < nz; ++i) {
        double s = 0;
        for (int j = 0; j < nz; ++j) s += M[i * nz + j] * W->b[j / 6].v[j % 6];
        u[i] = s + dt * W->b[i / 6].fext[i % 6];
    }
    int rc = 0;
    if (nc == 0) {
        const int n = nz + neq;
        double *P = calloc((size_t)n * n, sizeof(double));
        for (int i = 0; i < nz; ++i) for (int j = 0; j < nz; ++j) P[i * n + j] = M[i * nz + j];
        for (int e = 0; e < neq; ++e) for (int j = 0; j < nz; ++j) { P[j * n + nz + e] = -Je[e * nz + j]; P[(nz + e) * n + j] = Je[e * nz + j]; }
        rc = dense_solve(n, P, u);
        memcpy(newv, u, nz * sizeof(double));
        free(P);
    } else {
        const int nineq = nc * (fd + 2);
        double *G = calloc((size_t)nineq * nz, sizeof(double)), *F = calloc((size_t)nineq * nineq, sizeof(double)), *h = calloc(nineq, sizeof(double));
        double *b = calloc(neq + 1, sizeof(double));
        for (int c = 0; c < nc; ++c) {
            const contact_t *ct = &W->contacts.c[c];
            double x1[3], x2[3];
            cross3(ct->p1, ct->n, x1); cross3(ct->p2, ct->n, x2);
            double *row = G + (size_t)c * nz;
            for (int k = 0; k < 3; ++k) { row[6 * ct->b1 + k] = x1[k]; row[6 * ct->b1 + 3 + k] = ct->n[k]; }
            for (int k = 0; k < 3; ++k) { row[6 * ct->b2 + k] = -x2[k]; row[6 * ct->b2 + 3 + k] = -ct->n[k]; }   /* (b1 == b2 never happens) */
            double jv = 0;
            for (int j = 0; j < nz; ++j) jv += row[j] * W->b[j / 6].v[j % 6];
            h[c] = jv * ((W->b[ct->b1].rest + W->b[ct->b2].rest) / 2);
            /* friction directions */
            double dirs[8][3], o[3], t[3];
            orthogonal(ct->n, o); normalize3(o, dirs[0]);
            cross3(dirs[0], ct->n, t); normalize3(t, dirs[1]);
            int nd = 2;
            if (fd == 8) {
                for (int k = 0; k < 3; ++k) t[k] = dirs[0][k] + dirs[1][k];
                normalize3(t, dirs[2]);
                cross3(dirs[2], ct->n, t); normalize3(t, dirs[3]);
                nd = 4;
            }
            for (int d = 0; d < nd; ++d) for (int k = 0; k < 3; ++k) dirs[nd + d][k] = -dirs[d][k];
            for (int d = 0; d < 2 * nd; ++d) {
                double *fr = G + (size_t)(nc + c * fd + d) * nz;
                cross3(ct->p1, dirs[d], x1); cross3(ct->p2, dirs[d], x2);
                for (int k = 0; k < 3; ++k) { fr[6 * ct->b1 + k] = x1[k]; fr[6 * ct->b1 + 3 + k] = dirs[d][k]; }
                for (int k = 0; k < 3; ++k) { fr[6 * ct->b2 + k] = -x2[k]; fr[6 * ct->b2 + 3 + k] = -dirs[d][k]; }
                F[(size_t)(nc + c * fd + d) * nineq + (nc + nc * fd + c)] = 1.0;             /* E */
                F[(size_t)(nc + nc * fd + c) * nineq + (nc + c * fd + d)] = -1.0;            /* -E^T */
            }
            F[(size_t)(nc + nc * fd + c) * nineq + c] = 0.5 * (W->b[ct->b1].fric + W->b[ct->b2].fric);   /* mu */
        }
        double *z = calloc(nz, sizeof(double)), *lam = calloc(nineq, sizeof(double)), *sl = calloc(nineq, sizeof(double)), *nu = calloc(neq + 1, sizeof(double));
        int iters = 0; double best = 0;
        rc = lcp_oracle_forward1(M, u, G, h, Je, b, F, nz, nineq, neq, 1e-12, 3, W->max_iter, 1, z, lam, sl, nu, &iters, &best);
        W->n_lcp++; W->n_lcp_rows += nineq;
        for (int i = 0; i < nz; ++i) newv[i] = -z[i];
        if (W->lcp_bwd && rc == 0) {
            /* timing leg only: the implicit backward of this solve (lcp.py:156-213) with a unit upstream gradient; the
               reference runs it once per solve that stays on the autograd graph.  Gradients are discarded. */
            double *dQ = malloc(sizeof(double) * nz * nz), *dp = malloc(sizeof(double) * nz), *dG = malloc(sizeof(double) * nineq * nz);
            double *dh = malloc(sizeof(double) * nineq), *dA = malloc(sizeof(double) * (neq + 1) * nz), *db = malloc(sizeof(double) * (neq + 1));
            double *dF = malloc(sizeof(double) * (size_t)nineq * nineq), *one = malloc(sizeof(double) * nz);
            for (int i = 0; i < nz; ++i) one[i] = 1.0;
            lcp_oracle_backward1(M, G, Je, F, nz, nineq, neq, z, lam, sl, nu, one, dQ, dp, dG, dh, dA, db, dF);
            free(dQ); free(dp); free(dG); free(dh); free(dA); free(db); free(dF); free(one);
        }
        free(G); free(F); free(h); free(b); free(z); free(lam); free(sl); free(nu);
    }
    free(M); free(u); free(Je);
    return rc;
}

static void record_substep(world_t *W)
{
    if (W->nsub == W->sub_cap) {
        W->sub_cap = W->sub_cap ? 2 * W->sub_cap : 64;
        W->tr_t = realloc(W->tr_t, sizeof(double) * W->sub_cap);
        W->tr_p = realloc(W->tr_p, sizeof(double) * W->sub_cap * 7 * W->nb);
        W->tr_v = realloc(W->tr_v, sizeof(double) * W->sub_cap * 6 * W->nb);
        W->tr_nc = realloc(W->tr_nc, sizeof(int) * W->sub_cap);
        W->tr_c = realloc(W->tr_c, sizeof(clist_t) * W->sub_cap);
        for (int i = W->nsub; i < W->sub_cap; ++i) memset(&W->tr_c[i], 0, sizeof(clist_t));
    }
    const int k = W->nsub++;
    W->tr_t[k] = W->t;
    for (int i = 0; i < W->nb; ++i) { memcpy(W->tr_p + ((size_t)k * W->nb + i) * 7, W->b[i].p, 7 * sizeof(double)); memcpy(W->tr_v + ((size_t)k * W->nb + i) * 6, W->b[i].v, 6 * sizeof(double)); }
    W->tr_nc[k] = W->contacts.n;
    cl_copy(&W->tr_c[k], &W->contacts);
}

/* World.step_dt (world.py:241-379) */
static int step_dt(world_t *W, double dt)
{
    const int nb = W->nb;
    double *sp = malloc(sizeof(double) * 7 * nb), *sv = malloc(sizeof(double) * 6 * nb), *nv = malloc(sizeof(double) * 6 * nb);
    clist_t sc = {0};
    for (int i = 0; i < nb; ++i) { memcpy(sp + 7 * i, W->b[i].p, 7 * sizeof(double)); memcpy(sv + 6 * i, W->b[i].v, 6 * sizeof(double)); }
    cl_copy(&sc, &W->contacts);
    int rc = 0;
    while (1) {
        double dt_ = dt;
        if (W->toc_diff && W->have_toc) { double dtj = W->last_dt + dt_; dt_ = -W->last_dt + dtj; }
        W->n_attempts++;
        { const double t0 = now_s(); rc = solve_dynamics(W, dt_, nv); W->t_solve += now_s() - t0; }
        if (rc) break;
        for (int i = 0; i < nb; ++i) memcpy(W->b[i].v, nv + 6 * i, 6 * sizeof(double));
        for (int i = 0; i < nb; ++i) move_body(&W->b[i], dt_);
        find_contacts(W);
        if (W->err) { rc = W->err; break; }
        int ok = 1;
        for (int c = 0; c < W->contacts.n; ++c) if (!(W->contacts.c[c].pen <= W->tol)) ok = 0;
        if (ok) {
            /* toc_contacts: contacts of body pairs (unordered) that had no contact at the start of the step */
            int toc = 0;
            for (int c = 0; c < W->contacts.n && !toc; ++c) {
                int seen = 0;
                for (int p = 0; p < sc.n; ++p)
                    if ((sc.c[p].b1 == W->contacts.c[c].b1 && sc.c[p].b2 == W->contacts.c[c].b2) || (sc.c[p].b1 == W->contacts.c[c].b2 && sc.c[p].b2 == W->contacts.c[c].b1)) { seen = 1; break; }
                if (!seen) toc = 1;
            }
            W->have_toc = toc;
            if (W->toc_diff && toc) {
                /* H.apply is the identity on dt_ in the forward pass; the motion is undone and redone with it (world.py:323-341) */
                for (int i = 0; i < nb; ++i) set_p(&W->b[i], sp + 7 * i);
                for (int i = 0; i < nb; ++i) move_body(&W->b[i], dt_);
                W->last_dt = dt_;
            }
            break;
        }
        if (!W->strict && dt < W->dt / 1024.0) break;
        dt /= 2;
        for (int i = 0; i < nb; ++i) { set_p(&W->b[i], sp + 7 * i); memcpy(W->b[i].v, sv + 6 * i, 6 * sizeof(double)); }
        cl_copy(&W->contacts, &sc);
    }
    if (!rc) { record_substep(W); W->t += dt; }
    free(sp); free(sv); free(nv); free(sc.c);
    return rc;
}

/* ---- C API (ctypes) ------------------------------------------------------------------------------------------------------ */
void *so_world_create(int nb, const int *shape, const double *prm, const double *pose, const double *vel, const double *mass,
                      const double *inertia, const double *rest, const double *fric, const double *fext, const int *fixed,
                      const unsigned char *nocon, double dt, double eps, double tol, int fric_dirs, int strict, int toc_diff, int max_iter)
{
    world_t *W = calloc(1, sizeof(world_t));
    W->nb = nb; W->b = calloc(nb, sizeof(body_t)); W->nocon = malloc(nb * nb);
    memcpy(W->nocon, nocon, nb * nb);
    W->dt = dt; W->eps = eps; W->tol = tol; W->fric_dirs = fric_dirs; W->strict = strict; W->toc_diff = toc_diff; W->max_iter = max_iter;
    for (int i = 0; i < nb; ++i) {
        body_t *B = &W->b[i];
        B->shape = shape[i]; B->fixed = fixed[i];
        memcpy(B->prm, prm + 3 * i, 3 * sizeof(double));
        /* scale: bodies.py:781 (box: max(dims) * 1.5 / 2), :956 (sphere: rad * 1.5), :893 (cylinder: max(rad, height / 2) * 1.5) */
        if (B->shape == SH_BOX) B->scale = fmax2(fmax2(B->prm[0], B->prm[1]), B->prm[2]) * 1.5 / 2;
        else if (B->shape == SH_SPHERE) B->scale = B->prm[0] * 1.5;
        else B->scale = fmax2(B->prm[0], B->prm[1] / 2) * 1.5;
        memcpy(B->v, vel + 6 * i, 6 * sizeof(double));
        B->mass = mass[i]; memcpy(B->I, inertia + 9 * i, 9 * sizeof(double));
        B->rest = rest[i]; B->fric = fric[i]; memcpy(B->fext, fext + 6 * i, 6 * sizeof(double));
        set_p(B, pose + 7 * i);
    }
    return W;
}
void so_world_set_mesh(void *h, int body, const double *verts, int nv, const int *faces, int nf)
{
    body_t *B = &((world_t *)h)->b[body];
    B->verts = malloc(sizeof(double) * 3 * nv); memcpy(B->verts, verts, sizeof(double) * 3 * nv); B->nv = nv;
    B->faces = malloc(sizeof(int) * 3 * nf); memcpy(B->faces, faces, sizeof(int) * 3 * nf); B->nf = nf;
}
/* meshes shared between worlds (the 176 000-face floor of the benchmark scenes): borrowed, not copied, not freed */
void so_world_share_mesh(void *h, int body, const double *verts, int nv, const int *faces, int nf)
{
    body_t *B = &((world_t *)h)->b[body];
    B->verts = (double *)verts; B->nv = nv; B->faces = (int *)faces; B->nf = nf; B->borrowed = 1;
}
void so_set_hull_callback(void *h, hull_cb_t cb) { ((world_t *)h)->hull_cb = cb; }
void so_set_lcp_backward(void *h, int on) { ((world_t *)h)->lcp_bwd = on; }
/* World.__init__ (world.py:93-100): initial contacts; returns their number, or -1 on interpenetration with strict_no_pen */
int so_world_init(void *h)
{
    world_t *W = h;
    find_contacts(W);
    if (W->err) return -100 - W->err;
    if (W->strict) for (int c = 0; c < W->contacts.n; ++c) if (!(W->contacts.c[c].pen <= W->tol)) return -1;
    return W->contacts.n;
}
/* World.step(fixed_dt=True) (world.py:119-133) */
int so_world_step(void *h)
{
    world_t *W = h;
    const double end_t = W->t + W->dt;
    while (W->t < end_t) { int rc = step_dt(W, end_t - W->t); if (rc) return rc; }
    return 0;
}
/* n outer steps of several independent worlds, distributed over OpenMP threads (the timing leg) */
int so_worlds_run(void **hs, int nworlds, int nsteps, int nthreads)
{
    int worst = 0;
#pragma omp parallel for schedule(dynamic) num_threads(nthreads) reduction(max : worst)
    for (int w = 0; w < nworlds; ++w)
        for (int s = 0; s < nsteps; ++s) { int rc = so_world_step(hs[w]); if (rc > worst) worst = rc; if (rc) break; }
    return worst;
}
double so_world_time(void *h) { return ((world_t *)h)->t; }
int so_world_nsub(void *h) { return ((world_t *)h)->nsub; }
void so_world_counters(void *h, long *out) { world_t *W = h; out[0] = W->n_attempts; out[1] = W->n_lcp; out[2] = W->n_lcp_rows; out[3] = W->n_fw_cand; }
void so_world_state(void *h, double *pose, double *vel)
{
    world_t *W = h;
    for (int i = 0; i < W->nb; ++i) { memcpy(pose + 7 * i, W->b[i].p, 7 * sizeof(double)); memcpy(vel + 6 * i, W->b[i].v, 6 * sizeof(double)); }
}
static void export_contacts(const clist_t *L, int *body, double *geom, int *stable, double *lap)
{
    for (int c = 0; c < L->n; ++c) {
        const contact_t *ct = &L->c[c];
        if (body) { body[2 * c] = ct->b1; body[2 * c + 1] = ct->b2; }
        if (geom) { memcpy(geom + 10 * c, ct->n, 3 * sizeof(double)); memcpy(geom + 10 * c + 3, ct->p1, 3 * sizeof(double)); memcpy(geom + 10 * c + 6, ct->p2, 3 * sizeof(double)); geom[10 * c + 9] = ct->pen; }
        if (stable) stable[c] = ct->stable;
        if (lap) { lap[2 * c] = ct->lap[0]; lap[2 * c + 1] = ct->lap[1]; }
    }
}
void so_world_timers(void *h, double *out) { world_t *W = h; out[0] = W->t_solve; out[1] = W->t_detect; }
int so_world_ncontacts(void *h) { return ((world_t *)h)->contacts.n; }
void so_world_contacts(void *h, int *body, double *geom, int *stable, double *lap) { export_contacts(&((world_t *)h)->contacts, body, geom, stable, lap); }
/* trajectory record k (world.py:373-377): the time BEFORE the sub-step, the state and contacts AFTER it */
int so_world_substep(void *h, int k, double *t, double *pose, double *vel)
{
    world_t *W = h;
    if (k < 0 || k >= W->nsub) return -1;
    *t = W->tr_t[k];
    memcpy(pose, W->tr_p + (size_t)k * 7 * W->nb, sizeof(double) * 7 * W->nb);
    memcpy(vel, W->tr_v + (size_t)k * 6 * W->nb, sizeof(double) * 6 * W->nb);
    return W->tr_nc[k];
}
void so_world_substep_contacts(void *h, int k, int *body, double *geom, int *stable, double *lap) { export_contacts(&((world_t *)h)->tr_c[k], body, geom, stable, lap); }
void so_world_free(void *h)
{
    world_t *W = h;
    for (int i = 0; i < W->nb; ++i) if (!W->b[i].borrowed) { free(W->b[i].verts); free(W->b[i].faces); }
    for (int i = 0; i < W->sub_cap; ++i) free(W->tr_c[i].c);
    free(W->tr_t); free(W->tr_p); free(W->tr_v); free(W->tr_nc); free(W->tr_c); free(W->contacts.c); free(W->b); free(W->nocon); free(W);
}
