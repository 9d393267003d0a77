/*
 * lcp_oracle.c -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * Plain-C restatement of the reference's batched primal-dual interior-point LCP solver
 * and its implicit backward, following the reference CPU path step by step:
 *
 *   pre_factor_kkt   lcp_physics/lcp/solvers/batch.py:413-479
 *   factor_kkt       lcp_physics/lcp/solvers/batch.py:485-520  (partial-pivot LU of T = R + 1/d)
 *   solve_kkt        lcp_physics/lcp/solvers/batch.py:380-410
 *   forward          lcp_physics/lcp/solvers/batch.py:70-231   (Mehrotra predictor-corrector)
 *   get_step         lcp_physics/lcp/solvers/batch.py:234-237
 *   backward         lcp_physics/lcp/lcp.py:156-213
 *
 * One deliberate difference, stated in SURVEY.md §7: the reference's termination tests and
 * get_step's "max over the entire tensor" couple the systems of a batch; the reference's
 * engine always calls with a batch of one (engines.py:59-81), so this oracle applies them
 * per system.  For nBatch = 1 it is the same arithmetic.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * Pinned against tests/golden/lcp_*.npz, which were produced by the imported reference
 * (oracle/gen/gen_lcp_golden.py).
 *
 * All matrices row-major, double.  Returns 0 on success; 1 = Q not SPD
 * (lcp.py:109-113), 2 = LU of Q failed (batch.py:417-424).
 */
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdlib.h>
#include <string.h>

/* ---- dense helpers ------------------------------------------------------------------ */

/* Hot loops are compiled for several x86 vector widths and picked at load time (the library is built once and travels to
 * another host).  No FMA contraction anywhere (-ffp-contract=off), so every clone produces the same bits. */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define SO_CLONES __attribute__((target_clones("default", "avx2", "avx512f")))
#else
#define SO_CLONES
#endif

/* row[j] -= sum over the kb factor rows, one at a time in ascending order (the order of the unblocked elimination) */
SO_CLONES static void lu_row_update(int kb, int len, const double *l, double *row, const double *u, int ldu)
{
    for (int k = 0; k < kb; ++k) {
        const double lk = l[k];
        const double *uk = u + (size_t)k * ldu;
        if (lk != 0.0)
            for (int j = 0; j < len; ++j) row[j] -= lk * uk[j];
    }
}

/* In-place LU with partial pivoting (LAPACK getrf convention: piv[k] = row swapped with k).  Blocked for the cache (the
 * reference's LAPACK is): a panel of NB columns is eliminated, then every trailing row receives the panel's NB updates in one
 * pass.  Each element still sees the same subtractions in the same order as in the textbook elimination -- bit-identical. */
/* threads one factorisation may share its trailing update among (0 / 1: none -- the default, and what every timing uses; the
   tests that follow ONE scene over hundreds of steps switch it on to finish sooner) */
static int g_lu_threads = 0;
void lcp_oracle_set_lu_threads(int n) { g_lu_threads = n; }

static int lu_factor(int n, double *a, int *piv)
{
    enum { NB = 32 };
    int info = 0;
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int kb = n - k0 < NB ? n - k0 : NB, k1 = k0 + kb;
        for (int k = k0; k < k1; ++k) {
            int p = k;
            double best = fabs(a[k * n + k]);
            for (int i = k + 1; i < n; ++i) {
                double v = fabs(a[i * n + k]);
                if (v > best) { best = v; p = i; }
            }
            piv[k] = p;
            if (best == 0.0) { info = k + 1; continue; }
            if (p != k)
                for (int j = 0; j < n; ++j) { double t = a[k * n + j]; a[k * n + j] = a[p * n + j]; a[p * n + j] = t; }
            double inv = 1.0 / a[k * n + k];
            for (int i = k + 1; i < n; ++i) {
                double l = a[i * n + k] * inv;
                a[i * n + k] = l;
                if (l != 0.0) {
                    double *ri = a + i * n, *rk = a + k * n;
                    for (int j = k + 1; j < k1; ++j) ri[j] -= l * rk[j];      /* inside the panel only */
                }
            }
        }
        if (k1 == n) break;
        /* the panel's own rows, right of the panel: row i takes the updates of the panel rows above it */
        for (int i = k0 + 1; i < k1; ++i) lu_row_update(i - k0, n - k1, a + (size_t)i * n + k0, a + (size_t)i * n + k1, a + (size_t)k0 * n + k1, n);
        /* trailing rows: independent of each other (bit-identical in any order); shared among threads when the caller is not
           itself one of a team working on different systems */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_lu_threads > 1 ? g_lu_threads : 1) if (g_lu_threads > 1 && n - k1 >= 192 && !omp_in_parallel())
#endif
        for (int i = k1; i < n; ++i) lu_row_update(kb, n - k1, a + (size_t)i * n + k0, a + (size_t)i * n + k1, a + (size_t)k0 * n + k1, n);
    }
    return info;
}

/* Solve (P L U) x = b for nrhs right-hand sides stored as columns of b (n x nrhs row-major). */
static void lu_solve(int n, const double *lu, const int *piv, double *b, int nrhs)
{
    for (int k = 0; k < n; ++k)
        if (piv[k] != k)
            for (int c = 0; c < nrhs; ++c) { double t = b[k * nrhs + c]; b[k * nrhs + c] = b[piv[k] * nrhs + c]; b[piv[k] * nrhs + c] = t; }
    for (int i = 1; i < n; ++i)
        for (int k = 0; k < i; ++k) {
            double l = lu[i * n + k];
            if (l != 0.0) for (int c = 0; c < nrhs; ++c) b[i * nrhs + c] -= l * b[k * nrhs + c];
        }
    for (int i = n - 1; i >= 0; --i) {
        for (int k = i + 1; k < n; ++k) {
            double u = lu[i * n + k];
            if (u != 0.0) for (int c = 0; c < nrhs; ++c) b[i * nrhs + c] -= u * b[k * nrhs + c];
        }
        double inv = 1.0 / lu[i * n + i];
        for (int c = 0; c < nrhs; ++c) b[i * nrhs + c] *= inv;
    }
}

static void matvec(int m, int n, const double *a, const double *x, double *y)   /* y = A x */
{
    for (int i = 0; i < m; ++i) { double s = 0; for (int j = 0; j < n; ++j) s += a[i * n + j] * x[j]; y[i] = s; }
}
static void matvec_t(int m, int n, const double *a, const double *x, double *y) /* y = A^T x */
{
    for (int j = 0; j < n; ++j) y[j] = 0;
    for (int i = 0; i < m; ++i) { double xi = x[i]; for (int j = 0; j < n; ++j) y[j] += a[i * n + j] * xi; }
}
static double norm2(int n, const double *x) { double s = 0; for (int i = 0; i < n; ++i) s += x[i] * x[i]; return sqrt(s); }

/* SPD test standing in for "all eigenvalues of Q have positive real part" (lcp.py:109-113):
 * for the symmetric mass matrices the engine passes this is equivalent to Cholesky succeeding
 * on the symmetric part. */
static int is_spd(int n, const double *q, double *w)
{
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) w[i * n + j] = 0.5 * (q[i * n + j] + q[j * n + i]);
    for (int k = 0; k < n; ++k) {
        double d = w[k * n + k];
        for (int j = 0; j < k; ++j) d -= w[k * n + j] * w[k * n + j];
        if (!(d > 0.0)) return 0;
        d = sqrt(d); w[k * n + k] = d;
        for (int i = k + 1; i < n; ++i) {
            double s = w[i * n + k];
            for (int j = 0; j < k; ++j) s -= w[i * n + j] * w[k * n + j];
            w[i * n + k] = s / d;
        }
    }
    return 1;
}

/* ---- factorisation state ------------------------------------------------------------ */

typedef struct {
    int nz, nineq, neq;
    const double *Q, *G, *A;
    double *Q_LU; int *Q_piv;            /* LU(Q) */
    double *R;                           /* nineq x nineq */
    double *S_LU; int *S_piv;            /* (neq+nineq)^2 packed block LU of S */
    double *S21_base;                    /* G Q^-1 A^T U^-1 before row re-pivoting (nineq x neq) */
    double *T; int *T_piv;               /* scratch for LU(T) */
    double *tmp_nz, *tmp_nz2, *tmp_s;    /* scratch vectors */
} kkt_t;

static void kkt_free(kkt_t *k)
{
    free(k->Q_LU); free(k->Q_piv); free(k->R); free(k->S_LU); free(k->S_piv);
    free(k->S21_base); free(k->T); free(k->T_piv); free(k->tmp_nz); free(k->tmp_nz2); free(k->tmp_s);
}

/* batch.py:413-479 */
static int pre_factor_kkt(kkt_t *k, const double *Q, const double *G, const double *A, const double *F,
                          int nz, int nineq, int neq)
{
    int ns = neq + nineq;
    memset(k, 0, sizeof(*k));
    k->nz = nz; k->nineq = nineq; k->neq = neq; k->Q = Q; k->G = G; k->A = A;
    k->Q_LU = malloc(sizeof(double) * nz * nz); k->Q_piv = malloc(sizeof(int) * nz);
    k->R = malloc(sizeof(double) * (nineq ? nineq * nineq : 1));
    k->S_LU = calloc((size_t)ns * ns + 1, sizeof(double)); k->S_piv = malloc(sizeof(int) * (ns + 1));
    k->S21_base = malloc(sizeof(double) * (nineq * neq + 1));
    k->T = malloc(sizeof(double) * (nineq ? nineq * nineq : 1)); k->T_piv = malloc(sizeof(int) * (nineq + 1));
    k->tmp_nz = malloc(sizeof(double) * nz); k->tmp_nz2 = malloc(sizeof(double) * nz);
    k->tmp_s = malloc(sizeof(double) * (ns + 1));
    memcpy(k->Q_LU, Q, sizeof(double) * nz * nz);
    if (lu_factor(nz, k->Q_LU, k->Q_piv)) return 2;

    /* invQ_GT = Q^-1 G^T  (nz x nineq), R = G invQ_GT + F */
    double *invQ_GT = malloc(sizeof(double) * (nz * nineq + 1));
    for (int i = 0; i < nineq; ++i) for (int j = 0; j < nz; ++j) invQ_GT[j * nineq + i] = G[i * nz + j];
    lu_solve(nz, k->Q_LU, k->Q_piv, invQ_GT, nineq);
    for (int i = 0; i < nineq; ++i) {      /* (sum over l in ascending order, then + F: the order of the plain triple loop) */
        double *Ri = k->R + (size_t)i * nineq;
        for (int j = 0; j < nineq; ++j) Ri[j] = 0.0;
        for (int l = 0; l < nz; ++l) {
            const double g = G[i * nz + l], *X = invQ_GT + (size_t)l * nineq;
            for (int j = 0; j < nineq; ++j) Ri[j] += g * X[j];
        }
        for (int j = 0; j < nineq; ++j) Ri[j] += F[(size_t)i * nineq + j];
    }
    free(invQ_GT);
    for (int i = 0; i < ns; ++i) k->S_piv[i] = i;

    if (neq > 0) {
        double *invQ_AT = malloc(sizeof(double) * nz * neq);
        for (int i = 0; i < neq; ++i) for (int j = 0; j < nz; ++j) invQ_AT[j * neq + i] = A[i * nz + j];
        lu_solve(nz, k->Q_LU, k->Q_piv, invQ_AT, neq);
        double *AQA = malloc(sizeof(double) * neq * neq), *GQA = malloc(sizeof(double) * (nineq * neq + 1));
        for (int i = 0; i < neq; ++i) for (int j = 0; j < neq; ++j) {
            double s = 0; for (int l = 0; l < nz; ++l) s += A[i * nz + l] * invQ_AT[l * neq + j]; AQA[i * neq + j] = s; }
        for (int i = 0; i < nineq; ++i) for (int j = 0; j < neq; ++j) {
            double s = 0; for (int l = 0; l < nz; ++l) s += G[i * nz + l] * invQ_AT[l * neq + j]; GQA[i * neq + j] = s; }
        int *p11 = malloc(sizeof(int) * neq);
        lu_factor(neq, AQA, p11);                                   /* S_LU_11 = LU(A Q^-1 A^T) */
        /* S_LU_21 = GQA U^-1 : solve X U = GQA row by row (forward substitution on columns) */
        for (int r = 0; r < nineq; ++r)
            for (int j = 0; j < neq; ++j) {
                double s = GQA[r * neq + j];
                for (int l = 0; l < j; ++l) s -= k->S21_base[r * neq + l] * AQA[l * neq + j];
                k->S21_base[r * neq + j] = s / AQA[j * neq + j];
            }
        /* Tm = (A Q^-1 A^T)^-1 GQA^T (neq x nineq);  S_LU_12 = U Tm;  R -= GQA Tm */
        double *Tm = malloc(sizeof(double) * (neq * nineq + 1));
        for (int i = 0; i < nineq; ++i) for (int j = 0; j < neq; ++j) Tm[j * nineq + i] = GQA[i * neq + j];
        lu_solve(neq, AQA, p11, Tm, nineq);
        for (int i = 0; i < neq; ++i) {
            for (int j = 0; j < neq; ++j) k->S_LU[i * ns + j] = AQA[i * neq + j];
            for (int j = 0; j < nineq; ++j) {
                double s = 0; for (int l = i; l < neq; ++l) s += AQA[i * neq + l] * Tm[l * nineq + j];
                k->S_LU[i * ns + neq + j] = s;
            }
            k->S_piv[i] = p11[i];
        }
        for (int i = 0; i < nineq; ++i) for (int j = 0; j < nineq; ++j) {
            double s = 0; for (int l = 0; l < neq; ++l) s += GQA[i * neq + l] * Tm[l * nineq + j];
            k->R[i * nineq + j] -= s; }
        free(invQ_AT); free(AQA); free(GQA); free(p11); free(Tm);
    }
    return 0;
}

/* batch.py:485-520 */
static int factor_kkt(kkt_t *k, const double *d)
{
    int nineq = k->nineq, neq = k->neq, ns = neq + nineq;
    memcpy(k->T, k->R, sizeof(double) * nineq * nineq);
    for (int i = 0; i < nineq; ++i) k->T[i * nineq + i] += 1.0 / d[i];
    int info = lu_factor(nineq, k->T, k->T_piv);
    /* S_LU_21 <- P_new^T (G Q^-1 A^T U^-1): apply the new row interchanges in order */
    if (neq > 0) {
        double *w = malloc(sizeof(double) * nineq * neq);
        memcpy(w, k->S21_base, sizeof(double) * nineq * neq);
        for (int r = 0; r < nineq; ++r)
            if (k->T_piv[r] != r)
                for (int j = 0; j < neq; ++j) { double t = w[r * neq + j]; w[r * neq + j] = w[k->T_piv[r] * neq + j]; w[k->T_piv[r] * neq + j] = t; }
        for (int r = 0; r < nineq; ++r) for (int j = 0; j < neq; ++j) k->S_LU[(neq + r) * ns + j] = w[r * neq + j];
        free(w);
    }
    for (int r = 0; r < nineq; ++r) {
        k->S_piv[neq + r] = k->T_piv[r] + neq;
        for (int j = 0; j < nineq; ++j) k->S_LU[(neq + r) * ns + neq + j] = k->T[r * nineq + j];
    }
    return info;
}

/* batch.py:380-410.  Any of rx/rs/rz/ry may be NULL (= zeros). */
static void solve_kkt(kkt_t *k, const double *d, const double *rx, const double *rs, const double *rz,
                      const double *ry, double *dx, double *ds, double *dz, double *dy)
{
    int nz = k->nz, nineq = k->nineq, neq = k->neq, ns = neq + nineq;
    double *t = k->tmp_nz, *hh = k->tmp_s;
    for (int i = 0; i < nz; ++i) t[i] = rx ? rx[i] : 0.0;
    lu_solve(nz, k->Q_LU, k->Q_piv, t, 1);                       /* invQ_rx */
    if (neq) { matvec(neq, nz, k->A, t, hh); if (ry) for (int i = 0; i < neq; ++i) hh[i] -= ry[i]; }
    matvec(nineq, nz, k->G, t, hh + neq);
    for (int i = 0; i < nineq; ++i) hh[neq + i] += (rs ? rs[i] / d[i] : 0.0) - (rz ? rz[i] : 0.0);
    lu_solve(ns, k->S_LU, k->S_piv, hh, 1);
    for (int i = 0; i < ns; ++i) hh[i] = -hh[i];                  /* w */
    double *g1 = k->tmp_nz2;
    matvec_t(nineq, nz, k->G, hh + neq, g1);
    for (int i = 0; i < nz; ++i) g1[i] = -(rx ? rx[i] : 0.0) - g1[i];
    if (neq) { matvec_t(neq, nz, k->A, hh, t); for (int i = 0; i < nz; ++i) g1[i] -= t[i]; }
    lu_solve(nz, k->Q_LU, k->Q_piv, g1, 1);
    memcpy(dx, g1, sizeof(double) * nz);
    for (int i = 0; i < nineq; ++i) { dz[i] = hh[neq + i]; ds[i] = (-(rs ? rs[i] : 0.0) - hh[neq + i]) / d[i]; }
    if (neq && dy) memcpy(dy, hh, sizeof(double) * neq);
}

/* batch.py:234-237 (per system) */
static double get_step(int n, const double *v, const double *dv)
{
    double amax = -INFINITY;
    for (int i = 0; i < n; ++i) { double a = -v[i] / dv[i]; if (a > amax) amax = a; }
    double repl = amax > 1.0 ? amax : 1.0, amin = INFINITY;
    for (int i = 0; i < n; ++i) { double a = dv[i] > 0 ? repl : -v[i] / dv[i]; if (a < amin) amin = a; }
    return amin;
}

/* ---- public entry points (one system) ----------------------------------------------- */

int lcp_oracle_forward1(const double *Q, const double *p, const double *G, const double *h,
                        const double *A, const double *b, const double *F,
                        int nz, int nineq, int neq, double eps, int not_improved_lim, int max_iter,
                        int check_spd, double *zhat, double *lam, double *slack, double *nu,
                        int *iters, double *best_resid)
{
    kkt_t k;
    if (check_spd) {
        double *w = malloc(sizeof(double) * nz * nz);
        int ok = is_spd(nz, Q, w);
        free(w);
        if (!ok) return 1;
    }
    int rc = pre_factor_kkt(&k, Q, G, A, F, nz, nineq, neq);
    if (rc) { kkt_free(&k); return rc; }
    size_t nv = (size_t)nz + 3 * (size_t)nineq + neq;
    double *buf = calloc(16 * (nv + 8), sizeof(double)), *q = buf;
#define TAKE(n) (q += (n) + 1, q - (n) - 1)
    double *x = TAKE(nz), *s = TAKE(nineq), *z = TAKE(nineq), *y = TAKE(neq), *d = TAKE(nineq);
    double *rx = TAKE(nz), *rz = TAKE(nineq), *ry = TAKE(neq), *rs2 = TAKE(nineq), *negh = TAKE(nineq), *negb = TAKE(neq);
    double *dxa = TAKE(nz), *dsa = TAKE(nineq), *dza = TAKE(nineq), *dya = TAKE(neq);
    double *dxc = TAKE(nz), *dsc = TAKE(nineq), *dzc = TAKE(nineq), *dyc = TAKE(neq), *tn = TAKE(nz), *ti = TAKE(nineq);
#undef TAKE
    /* initial point: batch.py:85-110 */
    for (int i = 0; i < nineq; ++i) { d[i] = 1.0; negh[i] = -h[i]; }
    for (int i = 0; i < neq; ++i) negb[i] = -b[i];
    factor_kkt(&k, d);
    solve_kkt(&k, d, p, NULL, negh, neq ? negb : NULL, x, s, z, y);
    double m = INFINITY; for (int i = 0; i < nineq; ++i) if (s[i] < m) m = s[i];
    if (m < 0) for (int i = 0; i < nineq; ++i) s[i] -= m - 1;
    m = INFINITY; for (int i = 0; i < nineq; ++i) if (z[i] < m) m = z[i];
    if (m < 0) for (int i = 0; i < nineq; ++i) z[i] -= m - 1;

    double best = -1; int have_best = 0, not_improved = 0, it = 0;
    for (it = 0; it < max_iter; ++it) {
        /* residuals: batch.py:117-131 */
        matvec_t(nineq, nz, G, z, rx);
        matvec(nz, nz, Q, x, tn);
        for (int i = 0; i < nz; ++i) rx[i] += tn[i] + p[i];
        if (neq) { matvec_t(neq, nz, A, y, tn); for (int i = 0; i < nz; ++i) rx[i] += tn[i]; }
        matvec(nineq, nz, G, x, rz);
        matvec(nineq, nineq, F, z, ti);
        for (int i = 0; i < nineq; ++i) rz[i] += s[i] - h[i] - ti[i];
        if (neq) { matvec(neq, nz, A, x, ry); for (int i = 0; i < neq; ++i) ry[i] -= b[i]; }
        double sz = 0; for (int i = 0; i < nineq; ++i) sz += s[i] * z[i];
        double mu = fabs(sz / nineq);
        double resid = norm2(nineq, rz) + (neq ? norm2(neq, ry) : 0.0) + norm2(nz, rx) + nineq * mu;
        for (int i = 0; i < nineq; ++i) d[i] = z[i] / s[i];
        factor_kkt(&k, d);
        if (!have_best || resid < best) {
            best = resid; have_best = 1; not_improved = 0;
            memcpy(zhat, x, sizeof(double) * nz); memcpy(lam, z, sizeof(double) * nineq);
            memcpy(slack, s, sizeof(double) * nineq); if (neq) memcpy(nu, y, sizeof(double) * neq);
        } else {
            not_improved++;
        }
        if (not_improved == not_improved_lim || best < eps || mu > 1e32) break;
        /* affine direction: batch.py:174-192 */
        solve_kkt(&k, d, rx, z, rz, neq ? ry : NULL, dxa, dsa, dza, dya);
        double a1 = get_step(nineq, z, dza), a2 = get_step(nineq, s, dsa);
        double alpha = a1 < a2 ? a1 : a2; if (alpha > 1.0) alpha = 1.0;
        double t3 = 0;
        for (int i = 0; i < nineq; ++i) t3 += (s[i] + alpha * dsa[i]) * (z[i] + alpha * dza[i]);
        double sig = t3 / sz; sig = sig * sig * sig;
        for (int i = 0; i < nineq; ++i) rs2[i] = (-mu * sig + dsa[i] * dza[i]) / s[i];
        solve_kkt(&k, d, NULL, rs2, NULL, NULL, dxc, dsc, dzc, dyc);
        for (int i = 0; i < nz; ++i) dxa[i] += dxc[i];
        for (int i = 0; i < nineq; ++i) { dsa[i] += dsc[i]; dza[i] += dzc[i]; }
        for (int i = 0; i < neq; ++i) dya[i] += dyc[i];
        a1 = get_step(nineq, z, dza); a2 = get_step(nineq, s, dsa);
        alpha = 0.999 * (a1 < a2 ? a1 : a2); if (alpha > 1.0) alpha = 1.0;
        for (int i = 0; i < nz; ++i) x[i] += alpha * dxa[i];
        for (int i = 0; i < nineq; ++i) { s[i] += alpha * dsa[i]; z[i] += alpha * dza[i]; }
        for (int i = 0; i < neq; ++i) y[i] += alpha * dya[i];
    }
    if (iters) *iters = it;
    if (best_resid) *best_resid = best;
    free(buf); kkt_free(&k);
    return 0;
}

/* lcp.py:156-213.  Gradients have the shapes of the inputs; dA/db untouched when neq == 0. */
int lcp_oracle_backward1(const double *Q, const double *G, const double *A, const double *F,
                         int nz, int nineq, int neq,
                         const double *zhat, const double *lam, const double *slack, const double *nu,
                         const double *dl_dz,
                         double *dQ, double *dp, double *dG, double *dh, double *dA, double *db, double *dF)
{
    kkt_t k;
    int rc = pre_factor_kkt(&k, Q, G, A, F, nz, nineq, neq);
    if (rc) { kkt_free(&k); return rc; }
    double *d = malloc(sizeof(double) * (nineq + 1)), *dx = malloc(sizeof(double) * nz);
    double *ds = malloc(sizeof(double) * (nineq + 1)), *dlam = malloc(sizeof(double) * (nineq + 1));
    double *dnu = malloc(sizeof(double) * (neq + 1));
    for (int i = 0; i < nineq; ++i) {
        double l = lam[i] < 1e-8 ? 1e-8 : lam[i], s = slack[i] < 1e-8 ? 1e-8 : slack[i];
        d[i] = l / s;
    }
    factor_kkt(&k, d);
    solve_kkt(&k, d, dl_dz, NULL, NULL, NULL, dx, ds, dlam, dnu);
    for (int i = 0; i < nz; ++i) {
        dp[i] = dx[i];
        for (int j = 0; j < nz; ++j) dQ[i * nz + j] = 0.5 * (dx[i] * zhat[j] + zhat[i] * dx[j]);
    }
    for (int i = 0; i < nineq; ++i) {
        dh[i] = -dlam[i];
        for (int j = 0; j < nz; ++j) dG[i * nz + j] = dlam[i] * zhat[j] + lam[i] * dx[j];
        for (int j = 0; j < nineq; ++j) dF[i * nineq + j] = dlam[i] * lam[j];
    }
    for (int i = 0; i < neq; ++i) {
        db[i] = -dnu[i];
        for (int j = 0; j < nz; ++j) dA[i * nz + j] = dnu[i] * zhat[j] + nu[i] * dx[j];
    }
    free(d); free(dx); free(ds); free(dlam); free(dnu); kkt_free(&k);
    return 0;
}

/* ---- batched wrappers (OpenMP over independent systems when compiled with -fopenmp) --- */

int lcp_oracle_forward(const double *Q, const double *p, const double *G, const double *h,
                       const double *A, const double *b, const double *F,
                       int nbatch, int nz, int nineq, int neq, double eps, int not_improved_lim,
                       int max_iter, int check_spd, double *zhat, double *lam, double *slack, double *nu,
                       int *iters, int *status)
{
    int worst = 0;
#pragma omp parallel for schedule(dynamic) reduction(max : worst)
    for (int i = 0; i < nbatch; ++i) {
        double best = 0.0;
        int rc = lcp_oracle_forward1(Q + (size_t)i * nz * nz, p + (size_t)i * nz, G + (size_t)i * nineq * nz,
                                     h + (size_t)i * nineq, A + (size_t)i * neq * nz, b + (size_t)i * neq,
                                     F + (size_t)i * nineq * nineq, nz, nineq, neq, eps, not_improved_lim,
                                     max_iter, check_spd, zhat + (size_t)i * nz, lam + (size_t)i * nineq,
                                     slack + (size_t)i * nineq, nu + (size_t)i * neq,
                                     iters ? iters + i : NULL, &best);
        /* status 4: the best residual stayed above 1 -- where the reference prints INACC_ERR if verbose >= 0
           (batch.py:165-167, 229-230); the engine's verbose = -1 keeps it silent, the iterate is returned either way */
        if (rc == 0 && best > 1.0) rc = 4;
        if (status) status[i] = rc;
        if (rc > worst) worst = rc;
    }
    return worst;
}

int lcp_oracle_backward(const double *Q, const double *G, const double *A, const double *F,
                        int nbatch, int nz, int nineq, int neq,
                        const double *zhat, const double *lam, const double *slack, const double *nu,
                        const double *dl_dz, double *dQ, double *dp, double *dG, double *dh,
                        double *dA, double *db, double *dF)
{
    int worst = 0;
#pragma omp parallel for schedule(dynamic) reduction(max : worst)
    for (int i = 0; i < nbatch; ++i) {
        int rc = lcp_oracle_backward1(Q + (size_t)i * nz * nz, G + (size_t)i * nineq * nz, A + (size_t)i * neq * nz,
                                      F + (size_t)i * nineq * nineq, nz, nineq, neq, zhat + (size_t)i * nz,
                                      lam + (size_t)i * nineq, slack + (size_t)i * nineq, nu + (size_t)i * neq,
                                      dl_dz + (size_t)i * nz, dQ + (size_t)i * nz * nz, dp + (size_t)i * nz,
                                      dG + (size_t)i * nineq * nz, dh + (size_t)i * nineq,
                                      dA + (size_t)i * neq * nz, db + (size_t)i * neq, dF + (size_t)i * nineq * nineq);
        if (rc > worst) worst = rc;
    }
    return worst;
}
