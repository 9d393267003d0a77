// lcp_dense.hip -- general dense LCP (boundary B1) for gfx950: one wavefront per system.
//
// Replaces lcp_physics.lcp.lcp.LCPFunction (reference lcp.py:43-213) for ARBITRARY dense
// (Q,p,G,h,A,b,F): the same primal-dual interior point iteration as the reference
// (batch.py:70-231), with its dual-side block elimination (batch.py:380-520): LU(Q) once,
// R = G Q^-1 G^T + F - (G Q^-1 A^T)(A Q^-1 A^T)^-1(A Q^-1 G^T) once, and a partial-pivot LU
// of T = R + diag(s/z) every iteration.  The contact-structured fast path that the batched
// stepper uses lives in lcp_contact.hip; this kernel is the size- and structure-agnostic
// drop-in and the on-device cross-check for it.
//
// Mapping: blockDim = 64 (one wave), grid = B.  Matrices live in the caller's workspace
// (L2 / Infinity-Cache resident for the sizes of SURVEY.md §8: <= 2.6 MB per system),
// nz- and neq-sized vectors and the Schur right-hand side are staged in LDS, rows are
// walked with lane = column so every global access is a coalesced 512-B row segment.
#include "dss_device.h"
#include <math.h>
#include <stdlib.h>

#include "../../include/diffsdfsim_hip.h"
#include "wave_utils.h"

namespace {
using namespace dss;

struct Ws {
    double *QLU, *R, *T, *S11, *S12, *S21, *XG, *XA, *GQA, *Tm;
    double *s, *z, *d, *rz, *rs, *dsa, *dza, *dsc, *dzc, *ti;
    int *qpiv, *p11, *tpiv;
};

__host__ __device__ inline size_t ws_doubles(int nz, int nineq, int neq)
{
    size_t n = (size_t)nz * nz + 2 * (size_t)nineq * nineq + (size_t)neq * neq + 2 * (size_t)neq * nineq +
               (size_t)nz * nineq + (size_t)nz * neq + 2 * (size_t)nineq * neq + 10 * (size_t)nineq;
    return n + 8;
}
__host__ __device__ inline size_t ws_ints(int nz, int nineq, int neq) { return (size_t)nz + neq + nineq + 8; }
__host__ __device__ inline size_t ws_bytes_per_system(int nz, int nineq, int neq)
{
    size_t b = ws_doubles(nz, nineq, neq) * 8 + ws_ints(nz, nineq, neq) * 4;
    return (b + 255) & ~(size_t)255;
}

// staged vectors of a system (LDS), rounded to 16 bytes so that a workspace placed behind them stays aligned
__host__ __device__ inline size_t lds_bytes(int nz, int nineq, int neq)
{
    return (size_t)((neq + nineq + 12 * nz + 4 * (neq + 1) + 8 + 1) & ~1) * sizeof(double);
}
// workspace in LDS when vectors + workspace fit this budget (five systems per CU at the limit)
constexpr size_t WS_LDS_BUDGET = 30 * 1024;

__device__ inline Ws carve(char *base, int nz, int nineq, int neq)
{
    Ws w;
    double *q = reinterpret_cast<double *>(base);
    auto take = [&](size_t n) { double *r = q; q += n; return r; };
    w.QLU = take((size_t)nz * nz);
    w.R = take((size_t)nineq * nineq);
    w.T = take((size_t)nineq * nineq);
    w.S11 = take((size_t)neq * neq);
    w.S12 = take((size_t)neq * nineq);
    w.S21 = take((size_t)nineq * neq);
    w.XG = take((size_t)nz * nineq);
    w.XA = take((size_t)nz * neq);
    w.GQA = take((size_t)nineq * neq);
    w.Tm = take((size_t)neq * nineq);
    w.s = take(nineq); w.z = take(nineq); w.d = take(nineq); w.rz = take(nineq); w.rs = take(nineq);
    w.dsa = take(nineq); w.dza = take(nineq); w.dsc = take(nineq); w.dzc = take(nineq); w.ti = take(nineq);
    q += 8;
    int *ip = reinterpret_cast<int *>(q);
    w.qpiv = ip; ip += nz;
    w.p11 = ip; ip += neq;
    w.tpiv = ip;
    return w;
}

// ---- wave-cooperative dense kernels (all 64 lanes call; __syncthreads == wave fence) ------

// In-place LU with partial pivoting of the n x n row-major matrix a (leading dim lda).
// Returns 0, or k+1 if a zero pivot was met at step k (LAPACK getrf convention).
__device__ int lu_factor(int n, double *a, int lda, int *piv)
{
    const int lane = lane_id();
    int info = 0;
    for (int k = 0; k < n; ++k) {
        double best = -1.0;
        int bi = n;
        for (int i = k + lane; i < n; i += WAVE) {
            double v = fabs(a[(size_t)i * lda + k]);
            if (v > best) { best = v; bi = i; }
        }
        wave_argmax(best, bi);
        if (lane == 0) piv[k] = bi;
        if (!(best > 0.0)) { if (!info) info = k + 1; if (lane == 0) piv[k] = k; __syncthreads(); continue; }
        if (bi != k)
            for (int j = lane; j < n; j += WAVE) {
                double t = a[(size_t)k * lda + j];
                a[(size_t)k * lda + j] = a[(size_t)bi * lda + j];
                a[(size_t)bi * lda + j] = t;
            }
        __syncthreads();
        const double inv = 1.0 / a[(size_t)k * lda + k];
        // lanes walk the trailing columns; the multiplier of each row is recomputed by every lane
        for (int i = k + 1; i < n; ++i) {
            const double l = a[(size_t)i * lda + k] * inv;
            if (l != 0.0)
                for (int j = k + 1 + lane; j < n; j += WAVE) a[(size_t)i * lda + j] -= l * a[(size_t)k * lda + j];
        }
        __syncthreads();
        for (int i = k + 1 + lane; i < n; i += WAVE) a[(size_t)i * lda + k] *= inv;
        __syncthreads();
    }
    return info;
}

// Apply the row interchanges of piv to the vector x (LDS or global), in order.
__device__ void apply_piv(int n, const int *piv, double *x)
{
    if (lane_id() == 0)
        for (int k = 0; k < n; ++k) {
            int p = piv[k];
            if (p != k) { double t = x[k]; x[k] = x[p]; x[p] = t; }
        }
    __syncthreads();
}

// x <- L^-1 x (unit lower) then x <- U^-1 x, row oriented: lanes split the dot product of a row.
__device__ void lower_solve(int n, const double *lu, int lda, double *x)
{
    const int lane = lane_id();
    for (int i = 1; i < n; ++i) {
        double acc = 0.0;
        for (int k = lane; k < i; k += WAVE) acc += lu[(size_t)i * lda + k] * x[k];
        acc = wave_sum(acc);
        if (lane == 0) x[i] -= acc;
        __syncthreads();
    }
}
__device__ void upper_solve(int n, const double *lu, int lda, double *x)
{
    const int lane = lane_id();
    for (int i = n - 1; i >= 0; --i) {
        double acc = 0.0;
        for (int k = i + 1 + lane; k < n; k += WAVE) acc += lu[(size_t)i * lda + k] * x[k];
        acc = wave_sum(acc);
        if (lane == 0) x[i] = (x[i] - acc) / lu[(size_t)i * lda + i];
        __syncthreads();
    }
}
__device__ void lu_solve1(int n, const double *lu, int lda, const int *piv, double *x)
{
    apply_piv(n, piv, x);
    lower_solve(n, lu, lda, x);
    upper_solve(n, lu, lda, x);
}

// Multi right-hand side solve, B is n x nrhs row-major; one lane owns one column (no sync inside).
__device__ void lu_solve_cols(int n, const double *lu, int lda, const int *piv, double *b, int nrhs)
{
    for (int c = lane_id(); c < nrhs; c += WAVE) {
        for (int k = 0; k < n; ++k) {
            int p = piv[k];
            if (p != k) { double t = b[(size_t)k * nrhs + c]; b[(size_t)k * nrhs + c] = b[(size_t)p * nrhs + c]; b[(size_t)p * nrhs + c] = t; }
        }
        for (int i = 1; i < n; ++i) {
            double acc = b[(size_t)i * nrhs + c];
            for (int k = 0; k < i; ++k) acc -= lu[(size_t)i * lda + k] * b[(size_t)k * nrhs + c];
            b[(size_t)i * nrhs + c] = acc;
        }
        for (int i = n - 1; i >= 0; --i) {
            double acc = b[(size_t)i * nrhs + c];
            for (int k = i + 1; k < n; ++k) acc -= lu[(size_t)i * lda + k] * b[(size_t)k * nrhs + c];
            b[(size_t)i * nrhs + c] = acc / lu[(size_t)i * lda + i];
        }
    }
    __syncthreads();
}

// y[i] = sum_j a[i][j] x[j]  (rows sequential, lanes over columns; y may be LDS or global)
__device__ void matvec(int m, int n, const double *a, const double *x, double *y)
{
    const int lane = lane_id();
    for (int i = 0; i < m; ++i) {
        double acc = 0.0;
        for (int j = lane; j < n; j += WAVE) acc += a[(size_t)i * n + j] * x[j];
        acc = wave_sum(acc);
        if (lane == 0) y[i] = acc;
    }
    __syncthreads();
}
// y[j] = sum_i a[i][j] x[i]  (lanes over columns, coalesced; deterministic order over i)
__device__ void matvec_t(int m, int n, const double *a, const double *x, double *y)
{
    for (int j = lane_id(); j < n; j += WAVE) {
        double acc = 0.0;
        for (int i = 0; i < m; ++i) acc += a[(size_t)i * n + j] * x[i];
        y[j] = acc;
    }
    __syncthreads();
}
__device__ double wnorm2(int n, const double *x)
{
    double acc = 0.0;
    for (int i = lane_id(); i < n; i += WAVE) acc += x[i] * x[i];
    return sqrt(wave_sum(acc));
}

// Cholesky of the symmetric part of Q into scratch w: 1 if SPD (stands in for lcp.py:109-113).
__device__ int is_spd(int n, const double *q, double *w)
{
    const int lane = lane_id();
    for (int e = lane; e < n * n; e += WAVE) { int i = e / n, j = e % n; w[e] = 0.5 * (q[i * n + j] + q[j * n + i]); }
    __syncthreads();
    int ok = 1;
    for (int k = 0; k < n && ok; ++k) {
        double dk = w[k * n + k];
        if (!(dk > 0.0)) { ok = 0; break; }
        dk = sqrt(dk);
        __syncthreads();
        for (int i = k + lane; i < n; i += WAVE) w[i * n + k] = (i == k) ? dk : w[i * n + k] / dk;
        __syncthreads();
        for (int e = lane; e < (n - k - 1) * (n - k - 1); e += WAVE) {
            int i = k + 1 + e / (n - k - 1), j = k + 1 + e % (n - k - 1);
            if (j <= i) w[i * n + j] -= w[i * n + k] * w[j * n + k];
        }
        __syncthreads();
    }
    return ok;
}

struct Sys {
    int nz, nineq, neq;
    const double *Q, *G, *A, *F;
    Ws w;
    double *hs, *t1, *t2;  // LDS: Schur rhs (neq+nineq), two nz scratch vectors
};

// batch.py:413-479
__device__ int pre_factor(Sys &S)
{
    const int nz = S.nz, nineq = S.nineq, neq = S.neq, lane = lane_id();
    Ws &w = S.w;
    for (int e = lane; e < nz * nz; e += WAVE) w.QLU[e] = S.Q[e];
    __syncthreads();
    if (lu_factor(nz, w.QLU, nz, w.qpiv)) return DSS_LCP_Q_SINGULAR;
    // XG = Q^-1 G^T (nz x nineq)
    for (int i = 0; i < nineq; ++i)
        for (int j = lane; j < nz; j += WAVE) w.XG[(size_t)j * nineq + i] = S.G[(size_t)i * nz + j];
    __syncthreads();
    lu_solve_cols(nz, w.QLU, nz, w.qpiv, w.XG, nineq);
    // R = G XG + F
    for (int i = 0; i < nineq; ++i)
        for (int j = lane; j < nineq; j += WAVE) {
            double acc = 0.0;
            for (int l = 0; l < nz; ++l) acc += S.G[(size_t)i * nz + l] * w.XG[(size_t)l * nineq + j];
            w.R[(size_t)i * nineq + j] = acc + S.F[(size_t)i * nineq + j];
        }
    __syncthreads();
    if (neq > 0) {
        for (int i = 0; i < neq; ++i)
            for (int j = lane; j < nz; j += WAVE) w.XA[(size_t)j * neq + i] = S.A[(size_t)i * nz + j];
        __syncthreads();
        lu_solve_cols(nz, w.QLU, nz, w.qpiv, w.XA, neq);
        for (int e = lane; e < neq * neq; e += WAVE) {
            int i = e / neq, j = e % neq;
            double acc = 0.0;
            for (int l = 0; l < nz; ++l) acc += S.A[(size_t)i * nz + l] * w.XA[(size_t)l * neq + j];
            w.S11[e] = acc;
        }
        for (int e = lane; e < nineq * neq; e += WAVE) {
            int i = e / neq, j = e % neq;
            double acc = 0.0;
            for (int l = 0; l < nz; ++l) acc += S.G[(size_t)i * nz + l] * w.XA[(size_t)l * neq + j];
            w.GQA[e] = acc;
        }
        __syncthreads();
        lu_factor(neq, w.S11, neq, w.p11);
        // S21 = GQA U^-1 : one lane per row
        for (int r = lane; r < nineq; r += WAVE)
            for (int j = 0; j < neq; ++j) {
                double acc = w.GQA[(size_t)r * neq + j];
                for (int l = 0; l < j; ++l) acc -= w.S21[(size_t)r * neq + l] * w.S11[l * neq + j];
                w.S21[(size_t)r * neq + j] = acc / w.S11[j * neq + j];
            }
        // Tm = (A Q^-1 A^T)^-1 GQA^T
        for (int e = lane; e < nineq * neq; e += WAVE) { int i = e / neq, j = e % neq; w.Tm[(size_t)j * nineq + i] = w.GQA[e]; }
        __syncthreads();
        lu_solve_cols(neq, w.S11, neq, w.p11, w.Tm, nineq);
        for (int i = 0; i < neq; ++i)
            for (int j = lane; j < nineq; j += WAVE) {
                double acc = 0.0;
                for (int l = i; l < neq; ++l) acc += w.S11[i * neq + l] * w.Tm[(size_t)l * nineq + j];
                w.S12[(size_t)i * nineq + j] = acc;
            }
        for (int i = 0; i < nineq; ++i)
            for (int j = lane; j < nineq; j += WAVE) {
                double acc = 0.0;
                for (int l = 0; l < neq; ++l) acc += w.GQA[(size_t)i * neq + l] * w.Tm[(size_t)l * nineq + j];
                w.R[(size_t)i * nineq + j] -= acc;
            }
        __syncthreads();
    }
    return 0;
}

// batch.py:485-520 : T = R + diag(1/d), LU with partial pivoting
__device__ void factor_kkt(Sys &S, const double *d)
{
    const int nineq = S.nineq, lane = lane_id();
    Ws &w = S.w;
    for (int i = 0; i < nineq; ++i)
        for (int j = lane; j < nineq; j += WAVE) w.T[(size_t)i * nineq + j] = w.R[(size_t)i * nineq + j] + (i == j ? 1.0 / d[i] : 0.0);
    __syncthreads();
    lu_factor(nineq, w.T, nineq, w.tpiv);
}

// Solve S [wy; wz] = rhs in place in S.hs, S = P_S L_S U_S with the reference's block LU:
//   L_S = [[L11,0],[P_T^T S21, L_T]],  U_S = [[U11,S12],[0,U_T]],  P_S = diag(P11, P_T).
__device__ void schur_solve(Sys &S)
{
    const int nineq = S.nineq, neq = S.neq, lane = lane_id();
    Ws &w = S.w;
    double *h1 = S.hs, *h2 = S.hs + neq;
    if (neq > 0) {
        apply_piv(neq, w.p11, h1);
        lower_solve(neq, w.S11, neq, h1);
        for (int r = lane; r < nineq; r += WAVE) {
            double acc = 0.0;
            for (int l = 0; l < neq; ++l) acc += w.S21[(size_t)r * neq + l] * h1[l];
            h2[r] -= acc;
        }
        __syncthreads();
    }
    apply_piv(nineq, w.tpiv, h2);
    lower_solve(nineq, w.T, nineq, h2);
    upper_solve(nineq, w.T, nineq, h2);
    if (neq > 0) {
        for (int i = 0; i < neq; ++i) {
            double acc = 0.0;
            for (int j = lane; j < nineq; j += WAVE) acc += w.S12[(size_t)i * nineq + j] * h2[j];
            acc = wave_sum(acc);
            if (lane == 0) h1[i] -= acc;
        }
        __syncthreads();
        upper_solve(neq, w.S11, neq, h1);
    }
}

// batch.py:380-410.  rx/ry live in LDS (or are null = 0), rs/rz in global (or null).
__device__ void solve_kkt(Sys &S, const double *d, const double *rx, const double *rs, const double *rz,
                          const double *ry, double *dx, double *ds, double *dz, double *dy)
{
    const int nz = S.nz, nineq = S.nineq, neq = S.neq, lane = lane_id();
    Ws &w = S.w;
    double *t = S.t1, *g1 = S.t2, *hs = S.hs;
    for (int i = lane; i < nz; i += WAVE) t[i] = rx ? rx[i] : 0.0;
    __syncthreads();
    lu_solve1(nz, w.QLU, nz, w.qpiv, t);
    if (neq) {
        matvec(neq, nz, S.A, t, hs);
        if (ry) { for (int i = lane; i < neq; i += WAVE) hs[i] -= ry[i]; __syncthreads(); }
    }
    matvec(nineq, nz, S.G, t, hs + neq);
    for (int i = lane; i < nineq; i += WAVE) hs[neq + i] += (rs ? rs[i] / d[i] : 0.0) - (rz ? rz[i] : 0.0);
    __syncthreads();
    schur_solve(S);
    for (int i = lane; i < neq + nineq; i += WAVE) hs[i] = -hs[i];
    __syncthreads();
    matvec_t(nineq, nz, S.G, hs + neq, g1);
    for (int i = lane; i < nz; i += WAVE) g1[i] = -(rx ? rx[i] : 0.0) - g1[i];
    __syncthreads();
    if (neq) {
        matvec_t(neq, nz, S.A, hs, t);
        for (int i = lane; i < nz; i += WAVE) g1[i] -= t[i];
        __syncthreads();
    }
    lu_solve1(nz, w.QLU, nz, w.qpiv, g1);
    for (int i = lane; i < nz; i += WAVE) dx[i] = g1[i];
    for (int i = lane; i < nineq; i += WAVE) {
        double wz = hs[neq + i];
        dz[i] = wz;
        ds[i] = (-(rs ? rs[i] : 0.0) - wz) / d[i];
    }
    if (neq && dy) for (int i = lane; i < neq; i += WAVE) dy[i] = hs[i];
    __syncthreads();
}

// batch.py:234-237, per system
__device__ double get_step(int n, const double *v, const double *dv)
{
    double amax = -INFINITY;
    for (int i = lane_id(); i < n; i += WAVE) amax = fmax(amax, -v[i] / dv[i]);
    amax = wave_max(amax);
    const double repl = amax > 1.0 ? amax : 1.0;
    double amin = INFINITY;
    for (int i = lane_id(); i < n; i += WAVE) amin = fmin(amin, dv[i] > 0.0 ? repl : -v[i] / dv[i]);
    return wave_min(amin);
}

__device__ inline double *lds_take(double *&q, int n) { double *r = q; q += n; return r; }

__global__ void __launch_bounds__(64)
lcp_dense_forward_kernel(const double *Q, const double *p, const double *G, const double *h, const double *A,
                         const double *b, const double *F, int nz, int nineq, int neq, double eps,
                         int not_improved_lim, int max_iter, int check_spd, double *zhat, double *lam,
                         double *slack, double *nu, int *iters, int *status, char *workspace, size_t ws_stride, int ws_in_lds)
{
    DSS_DYN_LDS(double, lds);
    const int sys = blockIdx.x, lane = lane_id();
    Sys S;
    S.nz = nz; S.nineq = nineq; S.neq = neq;
    S.Q = Q + (size_t)sys * nz * nz; S.G = G + (size_t)sys * nineq * nz;
    S.A = neq ? A + (size_t)sys * neq * nz : nullptr; S.F = F + (size_t)sys * nineq * nineq;
    // small systems keep the whole workspace in LDS behind the staged vectors (ws_in_lds: chosen by the launcher): a
    // factorisation step is then an LDS round trip instead of an L2 / HBM one
    S.w = carve(ws_in_lds ? reinterpret_cast<char *>(lds) + lds_bytes(nz, nineq, neq) : workspace + (size_t)sys * ws_stride, nz, nineq, neq);
    p += (size_t)sys * nz; h += (size_t)sys * nineq; if (neq) b += (size_t)sys * neq;
    zhat += (size_t)sys * nz; lam += (size_t)sys * nineq; slack += (size_t)sys * nineq; if (neq) nu += (size_t)sys * neq;
    Ws &w = S.w;

    double *q = lds;
    S.hs = lds_take(q, neq + nineq); S.t1 = lds_take(q, nz); S.t2 = lds_take(q, nz);
    double *x = lds_take(q, nz), *y = lds_take(q, neq + 1), *rx = lds_take(q, nz), *ry = lds_take(q, neq + 1);
    double *dxa = lds_take(q, nz), *dya = lds_take(q, neq + 1), *dxc = lds_take(q, nz), *dyc = lds_take(q, neq + 1);
    double *pl = lds_take(q, nz), *tn = lds_take(q, nz);

    int st = DSS_LCP_OK;
    if (check_spd && !is_spd(nz, S.Q, w.QLU /* scratch, refilled by pre_factor */)) st = DSS_LCP_NOT_SPD;
    if (!st) st = pre_factor(S);
    if (st) {
        if (lane == 0) { status[sys] = st; iters[sys] = 0; }
        for (int i = lane; i < nz; i += WAVE) zhat[i] = 0.0;
        return;
    }
    double *s = w.s, *z = w.z, *d = w.d;
    // initial point, batch.py:85-110
    for (int i = lane; i < nineq; i += WAVE) { d[i] = 1.0; w.rz[i] = -h[i]; }
    for (int i = lane; i < nz; i += WAVE) pl[i] = p[i];
    for (int i = lane; i < neq; i += WAVE) ry[i] = -b[i];
    __syncthreads();
    factor_kkt(S, d);
    solve_kkt(S, d, pl, nullptr, w.rz, neq ? ry : nullptr, x, s, z, y);
    {
        double m = INFINITY;
        for (int i = lane; i < nineq; i += WAVE) m = fmin(m, s[i]);
        m = wave_min(m);
        if (m < 0) for (int i = lane; i < nineq; i += WAVE) s[i] -= m - 1.0;
        m = INFINITY;
        for (int i = lane; i < nineq; i += WAVE) m = fmin(m, z[i]);
        m = wave_min(m);
        if (m < 0) for (int i = lane; i < nineq; i += WAVE) z[i] -= m - 1.0;
        __syncthreads();
    }
    double best = 0.0;
    int have_best = 0, not_improved = 0, it = 0;
    for (it = 0; it < max_iter; ++it) {
        // residuals, batch.py:117-131
        matvec_t(nineq, nz, S.G, z, rx);
        matvec(nz, nz, S.Q, x, tn);
        for (int i = lane; i < nz; i += WAVE) rx[i] += tn[i] + pl[i];
        __syncthreads();
        if (neq) {
            matvec_t(neq, nz, S.A, y, tn);
            for (int i = lane; i < nz; i += WAVE) rx[i] += tn[i];
            __syncthreads();
        }
        matvec(nineq, nz, S.G, x, w.rz);
        matvec(nineq, nineq, S.F, z, w.ti);
        for (int i = lane; i < nineq; i += WAVE) w.rz[i] += s[i] - h[i] - w.ti[i];
        if (neq) { matvec(neq, nz, S.A, x, ry); for (int i = lane; i < neq; i += WAVE) ry[i] -= b[i]; }
        __syncthreads();
        double sz = 0.0;
        for (int i = lane; i < nineq; i += WAVE) sz += s[i] * z[i];
        sz = wave_sum(sz);
        const double mu = fabs(sz / nineq);
        const double resid = wnorm2(nineq, w.rz) + (neq ? wnorm2(neq, ry) : 0.0) + wnorm2(nz, rx) + nineq * mu;
        for (int i = lane; i < nineq; i += WAVE) d[i] = z[i] / s[i];
        __syncthreads();
        factor_kkt(S, d);
        if (!have_best || resid < best) {
            best = resid; have_best = 1; not_improved = 0;
            for (int i = lane; i < nz; i += WAVE) zhat[i] = x[i];
            for (int i = lane; i < nineq; i += WAVE) { lam[i] = z[i]; slack[i] = s[i]; }
            for (int i = lane; i < neq; i += WAVE) nu[i] = y[i];
        } else {
            ++not_improved;
        }
        if (not_improved == not_improved_lim || best < eps || mu > 1e32) break;
        // affine direction, batch.py:174-192
        solve_kkt(S, d, rx, z, w.rz, neq ? ry : nullptr, dxa, w.dsa, w.dza, dya);
        double alpha = fmin(fmin(get_step(nineq, z, w.dza), get_step(nineq, s, w.dsa)), 1.0);
        double t3 = 0.0;
        for (int i = lane; i < nineq; i += WAVE) t3 += (s[i] + alpha * w.dsa[i]) * (z[i] + alpha * w.dza[i]);
        t3 = wave_sum(t3);
        double sig = t3 / sz;
        sig = sig * sig * sig;
        for (int i = lane; i < nineq; i += WAVE) w.rs[i] = (-mu * sig + w.dsa[i] * w.dza[i]) / s[i];
        __syncthreads();
        solve_kkt(S, d, nullptr, w.rs, nullptr, nullptr, dxc, w.dsc, w.dzc, dyc);
        for (int i = lane; i < nz; i += WAVE) dxa[i] += dxc[i];
        for (int i = lane; i < nineq; i += WAVE) { w.dsa[i] += w.dsc[i]; w.dza[i] += w.dzc[i]; }
        for (int i = lane; i < neq; i += WAVE) dya[i] += dyc[i];
        __syncthreads();
        alpha = fmin(0.999 * fmin(get_step(nineq, z, w.dza), get_step(nineq, s, w.dsa)), 1.0);
        for (int i = lane; i < nz; i += WAVE) x[i] += alpha * dxa[i];
        for (int i = lane; i < nineq; i += WAVE) { s[i] += alpha * w.dsa[i]; z[i] += alpha * w.dza[i]; }
        for (int i = lane; i < neq; i += WAVE) y[i] += alpha * dya[i];
        __syncthreads();
    }
    if (lane == 0) { iters[sys] = it; status[sys] = (best > 1.0) ? DSS_LCP_INACCURATE : DSS_LCP_OK; }
}

// lcp.py:156-213
__global__ void __launch_bounds__(64)
lcp_dense_backward_kernel(const double *Q, const double *G, const double *A, const double *F, int nz, int nineq,
                          int neq, const double *zhat, const double *lam, const double *slack, const double *nu,
                          const double *dl_dz, double *dQ, double *dp, double *dG, double *dh, double *dA,
                          double *db, double *dF, char *workspace, size_t ws_stride, int ws_in_lds)
{
    DSS_DYN_LDS(double, lds);
    const int sys = blockIdx.x, lane = lane_id();
    Sys S;
    S.nz = nz; S.nineq = nineq; S.neq = neq;
    S.Q = Q + (size_t)sys * nz * nz; S.G = G + (size_t)sys * nineq * nz;
    S.A = neq ? A + (size_t)sys * neq * nz : nullptr; S.F = F + (size_t)sys * nineq * nineq;
    S.w = carve(ws_in_lds ? reinterpret_cast<char *>(lds) + lds_bytes(nz, nineq, neq) : workspace + (size_t)sys * ws_stride, nz, nineq, neq);
    Ws &w = S.w;
    zhat += (size_t)sys * nz; lam += (size_t)sys * nineq; slack += (size_t)sys * nineq; dl_dz += (size_t)sys * nz;
    dQ += (size_t)sys * nz * nz; dp += (size_t)sys * nz; dG += (size_t)sys * nineq * nz; dh += (size_t)sys * nineq;
    dF += (size_t)sys * nineq * nineq;
    if (neq) { nu += (size_t)sys * neq; dA += (size_t)sys * neq * nz; db += (size_t)sys * neq; }

    double *q = lds;
    S.hs = lds_take(q, neq + nineq); S.t1 = lds_take(q, nz); S.t2 = lds_take(q, nz);
    double *dx = lds_take(q, nz), *dnu = lds_take(q, neq + 1), *g = lds_take(q, nz), *zl = lds_take(q, nz);
    if (pre_factor(S)) return;
    for (int i = lane; i < nineq; i += WAVE) w.d[i] = fmax(lam[i], 1e-8) / fmax(slack[i], 1e-8);
    for (int i = lane; i < nz; i += WAVE) { g[i] = dl_dz[i]; zl[i] = zhat[i]; }
    __syncthreads();
    factor_kkt(S, w.d);
    solve_kkt(S, w.d, g, nullptr, nullptr, nullptr, dx, w.dsa, w.dza, dnu);
    const double *dlam = w.dza;
    for (int e = lane; e < nz * nz; e += WAVE) { int i = e / nz, j = e % nz; dQ[e] = 0.5 * (dx[i] * zl[j] + zl[i] * dx[j]); }
    for (int i = lane; i < nz; i += WAVE) dp[i] = dx[i];
    for (int i = lane; i < nineq; i += WAVE) dh[i] = -dlam[i];
    for (int i = 0; i < nineq; ++i) {
        const double dl = dlam[i], l = lam[i];
        for (int j = lane; j < nz; j += WAVE) dG[(size_t)i * nz + j] = dl * zl[j] + l * dx[j];
        for (int j = lane; j < nineq; j += WAVE) dF[(size_t)i * nineq + j] = dl * lam[j];
    }
    for (int i = 0; i < neq; ++i) {
        if (lane == 0) db[i] = -dnu[i];
        for (int j = lane; j < nz; j += WAVE) dA[(size_t)i * nz + j] = dnu[i] * zl[j] + nu[i] * dx[j];
    }
}

}  // namespace

namespace dss {      // lcp_dense_group.hip: eight lanes per system, for nz, nineq, neq <= 8
bool lcp_dense_group_fits(int nz, int nineq, int neq);
int launch_lcp_dense_group_forward(const double *Q, const double *p, const double *G, const double *h, const double *A, const double *b,
                                   const double *F, int B, int nz, int nineq, int neq, double eps, int not_improved_lim, int max_iter,
                                   int check_spd, double *zhat, double *lam, double *slack, double *nu, int *iters, int *status,
                                   hipStream_t stream);
int launch_lcp_dense_group_backward(const double *Q, const double *G, const double *A, const double *F, int B, int nz, int nineq, int neq,
                                    const double *zhat, const double *lam, const double *slack, const double *nu, const double *dl_dz,
                                    double *dQ, double *dp, double *dG, double *dh, double *dA, double *db, double *dF, hipStream_t stream);
}

extern "C" {

int dss_abi_version(void) { return DSS_ABI_VERSION; }

size_t dss_lcp_dense_workspace_bytes(int B, int nz, int nineq, int neq)
{
    if (B <= 0 || nz <= 0 || nineq < 0 || neq < 0) return 0;
    return (size_t)B * ws_bytes_per_system(nz, nineq, neq);
}

int dss_lcp_dense_forward(const double *Q, const double *p, const double *G, const double *h, const double *A,
                          const double *b, const double *F, int B, int nz, int nineq, int neq, double eps,
                          int not_improved_lim, int max_iter, int check_spd, double *zhat, double *lam,
                          double *slack, double *nu, int *iters, int *status, void *workspace,
                          size_t workspace_bytes, void *stream)
{
    if (B <= 0 || nz <= 0 || nineq <= 0 || neq < 0) return DSS_E_BADARG;
    if (!Q || !p || !G || !h || !F || !zhat || !lam || !slack || !iters || !status || !workspace) return DSS_E_BADARG;
    if (neq > 0 && (!A || !b || !nu)) return DSS_E_BADARG;
    if (workspace_bytes < dss_lcp_dense_workspace_bytes(B, nz, nineq, neq)) return DSS_E_WORKSPACE;
    // tiny systems: eight lanes per system, eight systems per wavefront, everything in registers (no workspace touched)
    if (dss::lcp_dense_group_fits(nz, nineq, neq) && !getenv("DSS_LCP_DENSE_WAVE"))
        return dss::launch_lcp_dense_group_forward(Q, p, G, h, A, b, F, B, nz, nineq, neq, eps, not_improved_lim, max_iter, check_spd, zhat,
                                                   lam, slack, nu, iters, status, (hipStream_t)stream);
    size_t lds = lds_bytes(nz, nineq, neq);
    if (lds > 64 * 1024) return DSS_E_UNSUPPORTED;
    const int in_lds = lds + ws_bytes_per_system(nz, nineq, neq) <= WS_LDS_BUDGET;
    if (in_lds) lds += ws_bytes_per_system(nz, nineq, neq);
    hipLaunchKernelGGL(lcp_dense_forward_kernel, dim3(B), dim3(64), lds, (hipStream_t)stream, Q, p, G, h, A, b, F,
                       nz, nineq, neq, eps, not_improved_lim, max_iter, check_spd, zhat, lam, slack, nu, iters,
                       status, (char *)workspace, ws_bytes_per_system(nz, nineq, neq), in_lds);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}

int dss_lcp_dense_backward(const double *Q, const double *G, const double *A, const double *F, int B, int nz,
                           int nineq, int neq, const double *zhat, const double *lam, const double *slack,
                           const double *nu, const double *dl_dz, double *dQ, double *dp, double *dG, double *dh,
                           double *dA, double *db, double *dF, void *workspace, size_t workspace_bytes,
                           void *stream)
{
    if (B <= 0 || nz <= 0 || nineq <= 0 || neq < 0) return DSS_E_BADARG;
    if (!Q || !G || !F || !zhat || !lam || !slack || !dl_dz || !dQ || !dp || !dG || !dh || !dF || !workspace) return DSS_E_BADARG;
    if (neq > 0 && (!A || !nu || !dA || !db)) return DSS_E_BADARG;
    if (workspace_bytes < dss_lcp_dense_workspace_bytes(B, nz, nineq, neq)) return DSS_E_WORKSPACE;
    if (dss::lcp_dense_group_fits(nz, nineq, neq) && !getenv("DSS_LCP_DENSE_WAVE"))
        return dss::launch_lcp_dense_group_backward(Q, G, A, F, B, nz, nineq, neq, zhat, lam, slack, nu, dl_dz, dQ, dp, dG, dh, dA, db, dF,
                                                    (hipStream_t)stream);
    size_t lds = lds_bytes(nz, nineq, neq);
    if (lds > 64 * 1024) return DSS_E_UNSUPPORTED;
    const int in_lds = lds + ws_bytes_per_system(nz, nineq, neq) <= WS_LDS_BUDGET;
    if (in_lds) lds += ws_bytes_per_system(nz, nineq, neq);
    hipLaunchKernelGGL(lcp_dense_backward_kernel, dim3(B), dim3(64), lds, (hipStream_t)stream, Q, G, A, F, nz, nineq,
                       neq, zhat, lam, slack, nu, dl_dz, dQ, dp, dG, dh, dA, db, dF, (char *)workspace,
                       ws_bytes_per_system(nz, nineq, neq), in_lds);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}

}  // extern "C"
