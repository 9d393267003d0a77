// lcp_dense_group.hip -- the general dense LCP (boundary B1) for batches of TINY systems: eight lanes per system.
//
// Same algorithm as lcp_dense.hip (the reference's primal-dual interior point iteration, batch.py:70-231, with its dual-side
// block elimination, batch.py:380-520), for nz, nineq, neq <= 8 -- BASELINE configs[0] (2-D ball on a slab: nz 6, nineq 4 or 8,
// neq 3) and anything else of that size.  lcp_dense.hip gives such a system a whole wavefront: every pivot search and every row of
// an 8 x 8 matrix is a 64-lane shuffle reduction with 8 live lanes, the matrices live in LDS, and 65 536 one-wave workgroups cost
// 6.5 ms in dispatch alone (0.1 % of the HBM roofline).  Here a wavefront carries EIGHT systems:
//   * lane r of a group owns ROW r of every matrix of its system (Q and its LU, G, G^T, A, A^T, F, R, T and its LU, the equality
//     blocks) in registers, all padded to 8 columns with compile-time indices, and ELEMENT r of every vector;
//   * a matrix-vector product is 8 broadcasts within the group (ds_bpermute) and 8 FMAs; the partial-pivot LU keeps rows in place
//     (an implicit permutation: the lane that wins the column's arg-max becomes the pivot and drops out) and broadcasts the pivot
//     row; triangular solves broadcast one solution component per step;
//   * operands are read straight from the caller's system-major arrays (a group's rows are consecutive: 8 systems of a wavefront read
//     one contiguous block per operand), nothing is staged in LDS, there is no workspace;
//   * the groups of a wavefront take different numbers of iterations: the loop runs until the last is done, a finished group
//     goes through the motions without committing anything (the shuffles need every lane).
// Rounding differs from lcp_dense.hip (another elimination order inside the equality block: it is solved as a 2 x 2 block system
// with its own LU of A Q^-1 A^T instead of the reference's spliced block LU); the Newton steps are the same, parity with the
// reference's goldens is held to 1e-9 like the wave-per-system kernel's.
#include <math.h>

#include "../../include/diffsdfsim_hip.h"
#include "wave_utils.h"

namespace {
using namespace dss;

constexpr int GS = 8;      // lanes per system = rows / columns of the register tiles

__device__ __forceinline__ int grp_lane() { return threadIdx.x & (GS - 1); }
// value of lane `src` of the own group, src a run-time number (the pivot lane of a pivoted elimination): ds_bpermute
__device__ __forceinline__ double gsh(double x, int src) { return __shfl(x, (threadIdx.x & (WAVE - GS)) | src, WAVE); }

// Cross-lane traffic with COMPILE-TIME partners runs on the DPP path of the vector ALU (a few cycles) instead of through the
// LDS crossbar (ds_bpermute / ds_swizzle: > 100 cycles each, and the solves are chains of them):
//   quad_perm for partners inside a quad, row_half_mirror (lane i <-> 7 - i of every 8) to cross between the two quads.
#if defined(DSS_EMU)
template <int CTRL> __device__ __forceinline__ double dpp_mov(double x)
{
    const int l = threadIdx.x & 63, q = l & ~3, i = l & 3;
    const int src = CTRL == 0x141 ? (l & ~7) | (7 - (l & 7)) : q | ((CTRL >> (2 * i)) & 3);
    return __shfl(x, src, WAVE);
}
template <int CTRL> __device__ __forceinline__ int dpp_movi(int x)
{
    const int l = threadIdx.x & 63, q = l & ~3, i = l & 3;
    const int src = CTRL == 0x141 ? (l & ~7) | (7 - (l & 7)) : q | ((CTRL >> (2 * i)) & 3);
    return __shfl(x, src, WAVE);
}
#else
template <int CTRL> __device__ __forceinline__ int dpp_movi(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ double dpp_mov(double x)
{
    return __hiloint2double(dpp_movi<CTRL>(__double2hiint(x)), dpp_movi<CTRL>(__double2loint(x)));
}
#endif
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141;     // quad_perm [1,0,3,2], [2,3,0,1]; row_half_mirror

// broadcast of lane J of every group of eight, J a compile-time constant.  gfx90a+ moves 64 bits per DPP instruction with the
// row_newbcast controls (lane N of every row of 16 to the whole row); a row holds two groups, told apart by the bank mask
// (banks 0-1 = lanes 0-7, banks 2-3 = lanes 8-15): two instructions per double.
template <int J> __device__ __forceinline__ double gbc(double x)
{
#if defined(DSS_EMU)
    return __shfl(x, (threadIdx.x & (WAVE - GS)) | J, WAVE);
#else
    const double lo = __builtin_amdgcn_update_dpp(x, x, 0x150 + J, 0xf, 0x3, false);          // lanes 8-15 of the row keep x for now
    return __builtin_amdgcn_update_dpp(lo, x, 0x150 + 8 + J, 0xf, 0xc, false);
#endif
}
__device__ __forceinline__ double gsum(double v)
{
    v += dpp_mov<DPP_XOR1>(v); v += dpp_mov<DPP_XOR2>(v); v += dpp_mov<DPP_HALF_MIRROR>(v);
    return v;
}
__device__ __forceinline__ double gmin(double v)
{
    v = fmin(v, dpp_mov<DPP_XOR1>(v)); v = fmin(v, dpp_mov<DPP_XOR2>(v)); v = fmin(v, dpp_mov<DPP_HALF_MIRROR>(v));
    return v;
}
__device__ __forceinline__ double gmax(double v)
{
    v = fmax(v, dpp_mov<DPP_XOR1>(v)); v = fmax(v, dpp_mov<DPP_XOR2>(v)); v = fmax(v, dpp_mov<DPP_HALF_MIRROR>(v));
    return v;
}
// arg-max over the group, lowest lane on ties (LAPACK idamax)
template <int CTRL> __device__ __forceinline__ void argmax_step(double &v, int &idx)
{
    const double ov = dpp_mov<CTRL>(v);
    const int oi = dpp_movi<CTRL>(idx);
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
}

// A row-distributed n x n matrix factored in place; rows never move.
//   PIV = true: partial pivoting by an implicit permutation -- the lane that wins the column's arg-max becomes the pivot of the
//     step and drops out; its row is broadcast with ds_bpermute (the source lane is a run-time number).  For T = R + diag(s/z),
//     which is not symmetric (F is not), as the reference factors it on the CPU (batch.py:33, 500-517).
//   PIV = false: natural order, every partner a compile-time lane (DPP).  For Q and A Q^-1 A^T, which are symmetric positive
//     definite: elimination without pivoting is backward stable there.
//   a[k] of a lane that was not yet pivot at step k = multiplier l_{r,k};  the step-k pivot lane holds U_{k,j} in a[j > k] and
//   1 / U_{k,k} in a[k] (one division per step of the factorisation instead of one per step of every solve).
struct GLU {
    double a[GS];
    int ord;          // the step at which this lane's row was the pivot (GS: never, i.e. r >= n)
    unsigned piv;     // pivot lane of step k in bits [3k, 3k+3)  (the same in every lane of the group)
};

template <bool PIV, int K> __device__ __forceinline__ void glu_factor_step(GLU &M, int n, int &info)
{
    if (K >= n) return;
    const int r = grp_lane();
    int p = K;
    double pr[GS];
    bool ok = true;
    if (PIV) {
        double v = (M.ord == GS && r < n) ? fabs(M.a[K]) : -1.0;
        int idx = r;
        argmax_step<DPP_XOR1>(v, idx); argmax_step<DPP_XOR2>(v, idx); argmax_step<DPP_HALF_MIRROR>(v, idx);
        p = idx;
        M.piv |= (unsigned)p << (3 * K);
        ok = v > 0.0;
#pragma unroll
        for (int j = K; j < GS; ++j) pr[j] = gsh(M.a[j], p);
    } else {
#pragma unroll
        for (int j = K; j < GS; ++j) pr[j] = gbc<K>(M.a[j]);
        ok = pr[K] != 0.0;
    }
    if (!ok && !info) info = K + 1;
    const double inv = 1.0 / pr[K];
    const bool is_pivot = r == p;
    const bool below = PIV ? (M.ord == GS && r < n && !is_pivot) : (r > K && r < n);
    if (is_pivot) { M.ord = K; M.a[K] = inv; }
    else if (below && ok) {
        const double l = M.a[K] * inv;
        M.a[K] = l;
#pragma unroll
        for (int j = K + 1; j < GS; ++j) M.a[j] -= l * pr[j];
    }
}
template <bool PIV> __device__ __forceinline__ int glu_factor(GLU &M, int n)
{
    M.ord = GS; M.piv = 0u;
    int info = 0;
    glu_factor_step<PIV, 0>(M, n, info); glu_factor_step<PIV, 1>(M, n, info); glu_factor_step<PIV, 2>(M, n, info);
    glu_factor_step<PIV, 3>(M, n, info); glu_factor_step<PIV, 4>(M, n, info); glu_factor_step<PIV, 5>(M, n, info);
    glu_factor_step<PIV, 6>(M, n, info); glu_factor_step<PIV, 7>(M, n, info);
    return info;
}

// Solve M X = B for NC right-hand sides at once: b[c] of lane r = B[r][c] (row r of B); on return b[c] of lane i = X[i][c].
template <bool PIV, int K, int NC> __device__ __forceinline__ void glu_fwd_step(const GLU &M, int n, double (&b)[NC])
{
    if (K >= n) return;
    const int r = grp_lane();
    const int p = PIV ? (int)((M.piv >> (3 * K)) & 7u) : K;
    const bool later = PIV ? (M.ord > K && r < n) : (r > K && r < n);     // rows that come later in the pivot order lose l * (pivot row's b)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double bp = PIV ? gsh(b[c], p) : gbc<K>(b[c]);
        if (later) b[c] -= M.a[K] * bp;
    }
}
template <bool PIV, int K, int NC> __device__ __forceinline__ void glu_bwd_step(const GLU &M, int n, double (&b)[NC], double (&x)[NC])
{
    if (K >= n) return;
    const int r = grp_lane();
    const int p = PIV ? (int)((M.piv >> (3 * K)) & 7u) : K;
    const bool earlier = PIV ? M.ord < K : r < K;          // component K is finished in the step-K pivot lane, then leaves the earlier rows
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double mine = b[c] * M.a[K];                  // (meaningful in the pivot lane only: a[K] = 1 / U_KK there)
        const double xk = PIV ? gsh(mine, p) : gbc<K>(mine);
        if (earlier) b[c] -= M.a[K] * xk;
        if (r == K) x[c] = xk;
    }
}
template <bool PIV, int NC> __device__ __forceinline__ void glu_solve(const GLU &M, int n, double (&b)[NC])
{
    glu_fwd_step<PIV, 0, NC>(M, n, b); glu_fwd_step<PIV, 1, NC>(M, n, b); glu_fwd_step<PIV, 2, NC>(M, n, b); glu_fwd_step<PIV, 3, NC>(M, n, b);
    glu_fwd_step<PIV, 4, NC>(M, n, b); glu_fwd_step<PIV, 5, NC>(M, n, b); glu_fwd_step<PIV, 6, NC>(M, n, b); glu_fwd_step<PIV, 7, NC>(M, n, b);
    double x[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) x[c] = 0.0;
    glu_bwd_step<PIV, 7, NC>(M, n, b, x); glu_bwd_step<PIV, 6, NC>(M, n, b, x); glu_bwd_step<PIV, 5, NC>(M, n, b, x); glu_bwd_step<PIV, 4, NC>(M, n, b, x);
    glu_bwd_step<PIV, 3, NC>(M, n, b, x); glu_bwd_step<PIV, 2, NC>(M, n, b, x); glu_bwd_step<PIV, 1, NC>(M, n, b, x); glu_bwd_step<PIV, 0, NC>(M, n, b, x);
#pragma unroll
    for (int c = 0; c < NC; ++c) b[c] = x[c];
}
template <bool PIV> __device__ __forceinline__ double glu_solve1(const GLU &M, int n, double b)
{
    double v[1] = {b};
    glu_solve<PIV, 1>(M, n, v);
    return v[0];
}

// y_r = sum_{j < n} m[j] x_j  (row r of a matrix times a vector whose element j lives in lane j)
template <int J> __device__ __forceinline__ void gmatvec_step(const double (&m)[GS], double x, int n, double &acc)
{
    if (J < n) acc += m[J] * gbc<J>(x);
}
__device__ __forceinline__ double gmatvec(const double (&m)[GS], double x, int n)
{
    double acc = 0.0;
    gmatvec_step<0>(m, x, n, acc); gmatvec_step<1>(m, x, n, acc); gmatvec_step<2>(m, x, n, acc); gmatvec_step<3>(m, x, n, acc);
    gmatvec_step<4>(m, x, n, acc); gmatvec_step<5>(m, x, n, acc); gmatvec_step<6>(m, x, n, acc); gmatvec_step<7>(m, x, n, acc);
    return acc;
}
// C = A B with A row-distributed (lane r: row r, k columns) and B row-distributed (lane l: row l): c[j] = sum_l a[l] B[l][j]
template <int L> __device__ __forceinline__ void gmatmul_step(const double (&a)[GS], const double (&b)[GS], int k, int ncol, double (&c)[GS])
{
    if (L >= k) return;
#pragma unroll
    for (int j = 0; j < GS; ++j) {
        if (j >= ncol) break;
        c[j] += a[L] * gbc<L>(b[j]);
    }
}
__device__ __forceinline__ void gmatmul(const double (&a)[GS], const double (&b)[GS], int k, int ncol, double (&c)[GS])
{
#pragma unroll
    for (int j = 0; j < GS; ++j) c[j] = 0.0;
    gmatmul_step<0>(a, b, k, ncol, c); gmatmul_step<1>(a, b, k, ncol, c); gmatmul_step<2>(a, b, k, ncol, c); gmatmul_step<3>(a, b, k, ncol, c);
    gmatmul_step<4>(a, b, k, ncol, c); gmatmul_step<5>(a, b, k, ncol, c); gmatmul_step<6>(a, b, k, ncol, c); gmatmul_step<7>(a, b, k, ncol, c);
}

struct GSys {
    int nz, ni, ne;
    double G[GS], GT[GS], A[GS], AT[GS], F[GS], Qr[GS];     // rows r of G (ni x nz), G^T (nz x ni), A (ne x nz), A^T (nz x ne), F (ni x ni), Q
    GLU Q, S11, T;                                            // LU(Q), LU(A Q^-1 A^T), LU(T) with T = R + diag(1/d)
    double Qi[GS], Si[GS];                                    // INV only: rows of Q^-1 and of (A Q^-1 A^T)^-1
    double R[GS], B12[GS], B21[GS];                           // R (ni x ni), A Q^-1 G^T (ne x ni), G Q^-1 A^T (ni x ne)
};

__device__ __forceinline__ void load_rows(double (&dst)[GS], const double *base, int nrow, int ncol, int rs, int cs)
{
    const int r = grp_lane();
#pragma unroll
    for (int j = 0; j < GS; ++j) dst[j] = (r < nrow && j < ncol) ? base[(size_t)r * rs + (size_t)j * cs] : 0.0;
}

// batch.py:413-479.  INV (the forward kernel): Q and A Q^-1 A^T are symmetric positive definite and fixed over the iterations, so
// their inverses are formed once (the LU solve of the identity) and every later solve with them is one matrix-vector product --
// eight independent broadcasts instead of a chain of sixteen dependent ones.  What the iterations lose in the last digits of a
// Newton direction they regain at the next residual, which is evaluated from the data; the adjoint (one solve, nothing to
// correct it afterwards) keeps the LU solves.
template <bool INV> __device__ __forceinline__ int g_pre_factor(GSys &S)
{
    const int nz = S.nz, ni = S.ni, ne = S.ne;
#pragma unroll
    for (int j = 0; j < GS; ++j) S.Q.a[j] = S.Qr[j];
    const int bad_q = glu_factor<false>(S.Q, nz);                    // (no early return: the groups of a wavefront stay in step)
    double XG[GS];                                            // Q^-1 G^T (nz x ni), row r
#pragma unroll
    for (int j = 0; j < GS; ++j) XG[j] = S.GT[j];
    glu_solve<false, GS>(S.Q, nz, XG);
    gmatmul(S.G, XG, nz, ni, S.R);                            // R = G Q^-1 G^T + F
#pragma unroll
    for (int j = 0; j < GS; ++j) S.R[j] += S.F[j];
    if (ne > 0) {
        double XA[GS], Tm[GS], t[GS];
#pragma unroll
        for (int j = 0; j < GS; ++j) XA[j] = S.AT[j];
        glu_solve<false, GS>(S.Q, nz, XA);                           // Q^-1 A^T (nz x ne)
        gmatmul(S.A, XA, nz, ne, S.S11.a);                    // A Q^-1 A^T
        gmatmul(S.G, XA, nz, ne, S.B21);                      // G Q^-1 A^T
        gmatmul(S.A, XG, nz, ni, S.B12);                      // A Q^-1 G^T
        glu_factor<false>(S.S11, ne);
#pragma unroll
        for (int j = 0; j < GS; ++j) Tm[j] = S.B12[j];
        glu_solve<false, GS>(S.S11, ne, Tm);                         // (A Q^-1 A^T)^-1 A Q^-1 G^T (ne x ni)
        gmatmul(S.B21, Tm, ne, ni, t);
#pragma unroll
        for (int j = 0; j < GS; ++j) S.R[j] -= t[j];
    }
    if (INV) {
        const int r = grp_lane();
#pragma unroll
        for (int j = 0; j < GS; ++j) { S.Qi[j] = j == r ? 1.0 : 0.0; S.Si[j] = j == r ? 1.0 : 0.0; }
        glu_solve<false, GS>(S.Q, nz, S.Qi);
        if (ne > 0) glu_solve<false, GS>(S.S11, ne, S.Si);
    }
    return bad_q ? DSS_LCP_Q_SINGULAR : 0;
}

// batch.py:485-520: T = R + diag(1/d), factored with partial pivoting as the reference factors it; afterwards the rows are MOVED
// into pivot order (lane k takes the row that was the pivot of step k: one ds_bpermute per register, not chained), which leaves a
// natural-order LU of P T: every solve with it then runs on compile-time partners, its right-hand side permuted on the way in.
__device__ __forceinline__ void g_factor_kkt(GSys &S, double d)
{
    const int r = grp_lane();
    const double id = 1.0 / d;
#pragma unroll
    for (int j = 0; j < GS; ++j) S.T.a[j] = S.R[j] + (j == r ? id : 0.0);
    glu_factor<true>(S.T, S.ni);
    const int src = r < S.ni ? (int)((S.T.piv >> (3 * r)) & 7u) : r;
    S.T.ord = src;                                           // (from here on: the lane whose right-hand-side element this lane takes)
#pragma unroll
    for (int j = 0; j < GS; ++j) S.T.a[j] = gsh(S.T.a[j], src);
}
__device__ __forceinline__ double g_solve_T(const GSys &S, double b) { return glu_solve1<false>(S.T, S.ni, gsh(b, S.T.ord)); }
template <bool INV> __device__ __forceinline__ double g_solve_Q(const GSys &S, double b)
{
    return INV ? gmatvec(S.Qi, b, S.nz) : glu_solve1<false>(S.Q, S.nz, b);
}
template <bool INV> __device__ __forceinline__ double g_solve_S11(const GSys &S, double b)
{
    return INV ? gmatvec(S.Si, b, S.ne) : glu_solve1<false>(S.S11, S.ne, b);
}

// batch.py:380-410; vectors have element i in lane i; has_* = 0 means the vector is zero
// RS_ONLY: rx = rz = ry = 0 (the corrector, batch.py:194-199): the products with those zeros are left out
template <bool INV, bool RS_ONLY = false>
__device__ __forceinline__ void g_solve_kkt(const GSys &S, double d, double rx, double rs, double rz, double ry, double &dx, double &ds,
                                            double &dz, double &dy)
{
    const int nz = S.nz, ni = S.ni, ne = S.ne, r = grp_lane();
    double h1 = 0.0, h2 = r < ni ? rs / d : 0.0;
    if (!RS_ONLY) {
        const double t = g_solve_Q<INV>(S, r < nz ? rx : 0.0);
        const double a2 = gmatvec(S.G, t, nz);
        h2 = r < ni ? a2 + rs / d - rz : 0.0;
        if (ne > 0) { const double a1 = gmatvec(S.A, t, nz); h1 = r < ne ? a1 - ry : 0.0; }
    }
    double w1 = 0.0, w2;
    if (ne > 0) {
        double h2p = h2;
        if (!RS_ONLY) { const double y1 = g_solve_S11<INV>(S, h1); h2p -= gmatvec(S.B21, y1, ne); }
        w2 = g_solve_T(S, r < ni ? h2p : 0.0);
        const double b12w = gmatvec(S.B12, w2, ni);          // (every lane takes part in the broadcasts, whatever it keeps)
        w1 = g_solve_S11<INV>(S, r < ne ? h1 - b12w : 0.0);
    } else {
        w2 = g_solve_T(S, h2);
    }
    w1 = -w1; w2 = -w2;
    double g1 = (RS_ONLY ? 0.0 : -rx) - gmatvec(S.GT, w2, ni);
    if (ne > 0) g1 -= gmatvec(S.AT, w1, ne);
    dx = g_solve_Q<INV>(S, r < nz ? g1 : 0.0);
    dz = r < ni ? w2 : 0.0;
    ds = r < ni ? (-rs - w2) / d : 0.0;
    dy = r < ne ? w1 : 0.0;
}

// batch.py:234-237, per system: element i in lane i (lanes >= n neutral)
__device__ __forceinline__ double g_get_step(double v, double dv, bool on)
{
    const double a = on ? -v / dv : -INFINITY;
    const double amax = gmax(a);
    const double repl = amax > 1.0 ? amax : 1.0;
    return gmin(on ? (dv > 0.0 ? repl : -v / dv) : INFINITY);
}

__device__ __forceinline__ void load_system(GSys &S, const double *Q, const double *G, const double *A, const double *F, size_t sys)
{
    const int nz = S.nz, ni = S.ni, ne = S.ne;
    load_rows(S.Qr, Q + sys * nz * nz, nz, nz, nz, 1);
    load_rows(S.G, G + sys * ni * nz, ni, nz, nz, 1);
    load_rows(S.GT, G + sys * ni * nz, nz, ni, 1, nz);
    load_rows(S.F, F + sys * ni * ni, ni, ni, ni, 1);
    if (ne > 0) {
        load_rows(S.A, A + sys * ne * nz, ne, nz, nz, 1);
        load_rows(S.AT, A + sys * ne * nz, nz, ne, 1, nz);
    } else {
#pragma unroll
        for (int j = 0; j < GS; ++j) { S.A[j] = 0.0; S.AT[j] = 0.0; }
    }
#pragma unroll
    for (int j = 0; j < GS; ++j) { S.B12[j] = 0.0; S.B21[j] = 0.0; S.S11.a[j] = 0.0; }
    S.S11.ord = GS; S.S11.piv = 0u;
}

template <int K> __device__ __forceinline__ void spd_step(double (&a)[GS], int nz, int &ok)
{
    if (K >= nz) return;
    const int r = grp_lane();
    double pr[GS];
#pragma unroll
    for (int j = K; j < GS; ++j) pr[j] = gbc<K>(a[j]);
    if (!(pr[K] > 0.0)) ok = 0;
    if (r > K && r < nz) {
        const double l = a[K] / pr[K];
#pragma unroll
        for (int j = K + 1; j < GS; ++j) a[j] -= l * pr[j];
    }
}
// lcp.py:109-113 stand-in: the symmetric part of Q is positive definite iff unpivoted elimination meets positive pivots only
__device__ __forceinline__ int g_is_spd(const GSys &S, const double *Q, size_t sys)
{
    const int nz = S.nz, r = grp_lane();
    double a[GS], qt[GS];
    load_rows(qt, Q + sys * nz * nz, nz, nz, 1, nz);
#pragma unroll
    for (int j = 0; j < GS; ++j) a[j] = 0.5 * (S.Qr[j] + qt[j]);
    int ok = 1;
    spd_step<0>(a, nz, ok); spd_step<1>(a, nz, ok); spd_step<2>(a, nz, ok); spd_step<3>(a, nz, ok);
    spd_step<4>(a, nz, ok); spd_step<5>(a, nz, ok); spd_step<6>(a, nz, ok); spd_step<7>(a, nz, ok);
    return ok;
}

#if !defined(DSS_GRP_WAVES)
#define DSS_GRP_WAVES 2
#endif
__global__ void __launch_bounds__(64, DSS_GRP_WAVES)
lcp_dense_group_forward_kernel(const double *Q, const double *p, const double *G, const double *h, const double *A, const double *b,
                               const double *F, int B, int nz, int ni, int ne, double eps, int not_improved_lim, int max_iter,
                               int check_spd, double *zhat, double *lam, double *slack, double *nu, int *iters, int *status)
{
    const int r = grp_lane();
    const long sys_raw = (long)blockIdx.x * (WAVE / GS) + (threadIdx.x >> 3);
    const bool real = sys_raw < B;
    const size_t sys = real ? (size_t)sys_raw : (size_t)(B - 1);      // a padding group repeats the last system and commits nothing
    GSys S;
    S.nz = nz; S.ni = ni; S.ne = ne;
    load_system(S, Q, G, A, F, sys);
    const double pv = r < nz ? p[sys * nz + r] : 0.0, hv = r < ni ? h[sys * ni + r] : 0.0, bv = (ne > 0 && r < ne) ? b[sys * ne + r] : 0.0;

    int st = DSS_LCP_OK;
    if (check_spd && !g_is_spd(S, Q, sys)) st = DSS_LCP_NOT_SPD;
    {
        const int rc = g_pre_factor<true>(S);       // (every group goes through it: the shuffles need all lanes)
        if (!st) st = rc;
    }
    // initial point, batch.py:85-110
    double x, s, z, y, d = 1.0;
    g_factor_kkt(S, d);
    g_solve_kkt<true>(S, d, pv, 0.0, -hv, -bv, x, s, z, y);
    {
        const bool on = r < ni;
        double m = gmin(on ? s : INFINITY);
        if (m < 0 && on) s -= m - 1.0;
        m = gmin(on ? z : INFINITY);
        if (m < 0 && on) z -= m - 1.0;
    }
    bool active = real && st == DSS_LCP_OK;
    double best = 0.0, bx = 0.0, bs = 0.0, bz = 0.0, by = 0.0;      // the best iterate so far stays in registers and is written once
    int have_best = 0, not_improved = 0, it_done = 0;
    for (int it = 0; it < max_iter; ++it) {
        if (__ballot(active) == 0ull) break;
        const bool on = r < ni;
        // residuals, batch.py:117-131
        double rx = gmatvec(S.GT, z, ni) + gmatvec(S.Qr, x, nz) + pv;
        if (ne > 0) rx += gmatvec(S.AT, y, ne);
        if (r >= nz) rx = 0.0;
        const double gx = gmatvec(S.G, x, nz), fz = gmatvec(S.F, z, ni);      // (every lane takes part in the broadcasts)
        const double rz = on ? gx + s - hv - fz : 0.0;
        double ry = 0.0;
        if (ne > 0) { const double ax = gmatvec(S.A, x, nz); ry = r < ne ? ax - bv : 0.0; }
        const double sz = gsum(on ? s * z : 0.0);
        const double mu = fabs(sz / ni);
        const double resid = sqrt(gsum(rz * rz)) + (ne > 0 ? sqrt(gsum(ry * ry)) : 0.0) + sqrt(gsum(rx * rx)) + ni * mu;
        d = on ? z / s : 1.0;
        g_factor_kkt(S, d);
        if (active) {
            if (!have_best || resid < best) {
                best = resid; have_best = 1; not_improved = 0;
                bx = x; bs = s; bz = z; by = y;
            } else {
                ++not_improved;
            }
            if (not_improved == not_improved_lim || best < eps || mu > 1e32) { active = false; it_done = it; }
        }
        // affine direction, batch.py:174-192
        double dxa, dsa, dza, dya, dxc, dsc, dzc, dyc;
        g_solve_kkt<true>(S, d, rx, on ? z : 0.0, rz, ry, dxa, dsa, dza, dya);
        double alpha = fmin(fmin(g_get_step(z, dza, on), g_get_step(s, dsa, on)), 1.0);
        const double t3 = gsum(on ? (s + alpha * dsa) * (z + alpha * dza) : 0.0);
        double sig = t3 / sz;
        sig = sig * sig * sig;
        const double rs2 = on ? (-mu * sig + dsa * dza) / s : 0.0;
        g_solve_kkt<true, true>(S, d, 0.0, rs2, 0.0, 0.0, dxc, dsc, dzc, dyc);
        dxa += dxc; dsa += dsc; dza += dzc; dya += dyc;
        alpha = fmin(0.999 * fmin(g_get_step(z, dza, on), g_get_step(s, dsa, on)), 1.0);
        if (active) {
            if (r < nz) x += alpha * dxa;
            if (on) { s += alpha * dsa; z += alpha * dza; }
            if (r < ne) y += alpha * dya;
            it_done = it + 1;
        }
    }
    if (real && r == 0) {
        if (st != DSS_LCP_OK) { status[sys] = st; iters[sys] = 0; }
        else { iters[sys] = it_done; status[sys] = (best > 1.0) ? DSS_LCP_INACCURATE : DSS_LCP_OK; }
    }
    if (real && st == DSS_LCP_OK && have_best) {
        if (r < nz) zhat[sys * nz + r] = bx;
        if (r < ni) { lam[sys * ni + r] = bz; slack[sys * ni + r] = bs; }
        if (ne > 0 && r < ne) nu[sys * ne + r] = by;
    }
    if (real && st != DSS_LCP_OK && r < nz) zhat[sys * nz + r] = 0.0;
}

// lcp.py:156-213
__global__ void __launch_bounds__(64)
lcp_dense_group_backward_kernel(const double *Q, const double *G, const double *A, const double *F, int B, int nz, int ni, int ne,
                                const double *zhat, const double *lam, const double *slack, const double *nu, const double *dl_dz,
                                double *dQ, double *dp, double *dG, double *dh, double *dA, double *db, double *dF)
{
    const int r = grp_lane();
    const long sys_raw = (long)blockIdx.x * (WAVE / GS) + (threadIdx.x >> 3);
    const bool real = sys_raw < B;
    const size_t sys = real ? (size_t)sys_raw : (size_t)(B - 1);
    GSys S;
    S.nz = nz; S.ni = ni; S.ne = ne;
    load_system(S, Q, G, A, F, sys);
    const int rc = g_pre_factor<false>(S);
    const bool on = r < ni;
    const double lv = on ? lam[sys * ni + r] : 1.0, sv = on ? slack[sys * ni + r] : 1.0;
    const double d = on ? fmax(lv, 1e-8) / fmax(sv, 1e-8) : 1.0;
    g_factor_kkt(S, d);
    const double g = r < nz ? dl_dz[sys * nz + r] : 0.0, zl = r < nz ? zhat[sys * nz + r] : 0.0;
    const double nv = (ne > 0 && r < ne) ? nu[sys * ne + r] : 0.0;
    double dx, dsv, dlam, dnu;
    g_solve_kkt<false>(S, d, g, 0.0, 0.0, 0.0, dx, dsv, dlam, dnu);
    const bool w = real && rc == 0;
    // outer products: row r of each gradient needs the whole of z / dx / lam
    const double lz = on ? lv : 0.0;
#define DSS_GRP_OUTER(J)                                                                                                 \
    {                                                                                                                    \
        const double zj = gbc<J>(zl), dxj = gbc<J>(dx), lj = gbc<J>(lz);                                                 \
        if (w && J < nz) {                                                                                               \
            if (r < nz) dQ[sys * nz * nz + (size_t)r * nz + J] = 0.5 * (dx * zj + zl * dxj);                              \
            if (on) dG[sys * ni * nz + (size_t)r * nz + J] = dlam * zj + lv * dxj;                                        \
            if (ne > 0 && r < ne) dA[sys * ne * nz + (size_t)r * nz + J] = dnu * zj + nv * dxj;                           \
        }                                                                                                                \
        if (w && J < ni && on) dF[sys * ni * ni + (size_t)r * ni + J] = dlam * lj;                                        \
    }
    DSS_GRP_OUTER(0) DSS_GRP_OUTER(1) DSS_GRP_OUTER(2) DSS_GRP_OUTER(3) DSS_GRP_OUTER(4) DSS_GRP_OUTER(5) DSS_GRP_OUTER(6) DSS_GRP_OUTER(7)
#undef DSS_GRP_OUTER
    if (w) {
        if (r < nz) dp[sys * nz + r] = dx;
        if (on) dh[sys * ni + r] = -dlam;
        if (ne > 0 && r < ne) db[sys * ne + r] = -dnu;
    }
}

}  // namespace

namespace dss {
bool lcp_dense_group_fits(int nz, int nineq, int neq) { return nz <= GS && nineq <= GS && neq <= GS; }

int launch_lcp_dense_group_forward(const double *Q, const double *p, const double *G, const double *h, const double *A, const double *b,
                                   const double *F, int B, int nz, int nineq, int neq, double eps, int not_improved_lim, int max_iter,
                                   int check_spd, double *zhat, double *lam, double *slack, double *nu, int *iters, int *status,
                                   hipStream_t stream)
{
    const int per = WAVE / GS, grid = (B + per - 1) / per;
    hipLaunchKernelGGL(lcp_dense_group_forward_kernel, dim3(grid), dim3(WAVE), 0, stream, Q, p, G, h, A, b, F, B, nz, nineq, neq, eps,
                       not_improved_lim, max_iter, check_spd, zhat, lam, slack, nu, iters, status);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}

int launch_lcp_dense_group_backward(const double *Q, const double *G, const double *A, const double *F, int B, int nz, int nineq, int neq,
                                    const double *zhat, const double *lam, const double *slack, const double *nu, const double *dl_dz,
                                    double *dQ, double *dp, double *dG, double *dh, double *dA, double *db, double *dF, hipStream_t stream)
{
    const int per = WAVE / GS, grid = (B + per - 1) / per;
    hipLaunchKernelGGL(lcp_dense_group_backward_kernel, dim3(grid), dim3(WAVE), 0, stream, Q, G, A, F, B, nz, nineq, neq, zhat, lam, slack,
                       nu, dl_dz, dQ, dp, dG, dh, dA, db, dF);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}
}  // namespace dss
