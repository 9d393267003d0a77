// lcp_contact.hip -- contact-structured frictional LCP for gfx950: one wavefront per scene, the reduced KKT
// system factored and solved in registers.
//
// Solves exactly the mixed LCP the reference's PdipmEngine assembles (engines.py:56-81) with the
// same predictor-corrector iteration as the reference solver (batch.py:70-231), but never builds
// the dense G [nineq x nz] and F [nineq x nineq]:
//
//   * every inequality row of a contact c between bodies (b1,b2) is  +-[p1 x D, D | -p2 x D, -D]
//     for a direction D in {n, d_1..d_ND} (physics3d/world.py:56-101), i.e. row = (P_c D)^T with
//     P_c = [X(p1); I; -X(p2); -I] (12x3);
//   * F couples only the (2 ND + 2) rows of one contact (engines.py:72-78), so (F + diag(s/z)) is
//     block diagonal with an arrow-shaped block per contact that is inverted in closed form.
//
// Eliminating ds and dz contact by contact (instead of the reference's dual-side Schur complement
// T = R + D^-1 of size nineq, batch.py:485-520) leaves the reduced system
//        [ Q + sum_c P_c C_c P_c^T   A^T ] [dx]   [rhs_x]
//        [ A                          0  ] [dy] = [rhs_y]          (n = nz + neq <= 64)
// with a 3x3 matrix C_c per contact.  The two are the same Newton system, so the iterates agree
// with the reference to rounding (verified against its goldens: 1e-13 on the velocities).
//
// Mapping: blockDim = 64 (one wave) per scene.  LDS: H = Q + sum P C P^T (odd leading dimension => conflict free
// column walks), contact points, per-contact 3-vectors / C matrices, nz-sized vectors.  The compiled system sizes
// (n = 54: 8 bodies + 6 equality rows; n = 18) are factored with one matrix row per lane in registers (kkt_reg.h);
// a body pinned by identity equality rows is eliminated in closed form first.  With maxc <= 128 the per-contact IPM
// state (s, z, rz, ds, dz, d: NR doubles each) of two contacts per lane lives in registers
// (lcp_contact_forward_reg_kernel); beyond that it streams through an L2-resident workspace in [row][contact] order
// (lcp_contact_forward_kernel); any other n <= 64 falls back to an LU in LDS.  Sums over contacts (K = Q + sum P C P^T, the
// gathers G^T u) are formed by lanes that each own a piece of a (body1, body2) run of contacts and add their partial sums with
// LDS atomics -- one wavefront, one instruction stream, so the order of the additions is fixed: results are bitwise
// reproducible run to run (tested at B = 1024), though not in contact order any more.
#include <math.h>

#include "../../include/diffsdfsim_hip.h"
#include "kkt_reg.h"
#include "wave_utils.h"

#if defined(DSS_DIAG)   // diagnostic build only (tools/lcp_phases.py); the product library has no global state
__device__ long long *g_lcp_stamps = nullptr;
#define LSTAMP(i) do { if (g_lcp_stamps && threadIdx.x == 0) atomicAdd((unsigned long long *)&g_lcp_stamps[(size_t)blockIdx.x * 16 + (i)], (unsigned long long)(wall_clock64() - t_last)); t_last = wall_clock64(); } while (0)
#define LSTAMP_INIT long long t_last = wall_clock64()
#else
#define LSTAMP(i) do { } while (0)
#define LSTAMP_INIT do { } while (0)
#endif

namespace {
using namespace dss;

// The contact index of a pass, hidden from loop-invariant code motion: otherwise every per-contact address of
// every pass (hundreds of 64-bit values) is hoisted out of the IPM iteration loop and spilled to scratch memory.
__device__ __forceinline__ int opaque_lane(int lane) { return dss_opaque(lane); }

template <int ND> struct Geo {
    static constexpr int NR = 2 * ND + 2;       // rows per contact: normal, ND +dirs, ND -dirs, cone
    static constexpr int NF = 3 * (1 + ND) + 8; // operand fields per contact
    double D[1 + ND][3];
    double p1[3], p2[3], mu, hn;
    int b1, b2;
};

template <int ND> __device__ inline void load_geo(Geo<ND> &g, const double *cop, const int *cbody, int maxc, int c)
{
#pragma unroll
    for (int k = 0; k < 1 + ND; ++k)
#pragma unroll
        for (int j = 0; j < 3; ++j) g.D[k][j] = cop[(size_t)(3 * k + j) * maxc + c];
    const int o = 3 * (1 + ND);
#pragma unroll
    for (int j = 0; j < 3; ++j) { g.p1[j] = cop[(size_t)(o + j) * maxc + c]; g.p2[j] = cop[(size_t)(o + 3 + j) * maxc + c]; }
    g.mu = cop[(size_t)(o + 6) * maxc + c];
    g.hn = cop[(size_t)(o + 7) * maxc + c];
    g.b1 = cbody[c];
    g.b2 = cbody[maxc + c];
}

__device__ inline void cross3(const double *a, const double *b, double *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ inline double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// relative velocity functional of a generalized vector v (LDS, [w(3), u(3)] per body):
//   vr = (w1 x p1 + u1) - (w2 x p2 + u2);   (G v)_row = +-D . vr
template <int ND> __device__ inline void rel_vel(const double *v, const Geo<ND> &g, double *vr)
{
    const double *a = v + 6 * g.b1, *b = v + 6 * g.b2;
    double t1[3], t2[3];
    cross3(a, g.p1, t1);
    cross3(b, g.p2, t2);
#pragma unroll
    for (int j = 0; j < 3; ++j) vr[j] = (t1[j] + a[3 + j]) - (t2[j] + b[3 + j]);
}
template <int ND> __device__ inline void g_rows(const Geo<ND> &g, const double *vr, double *gx)
{
    gx[0] = dot3(g.D[0], vr);
#pragma unroll
    for (int k = 1; k <= ND; ++k) { double t = dot3(g.D[k], vr); gx[k] = t; gx[ND + k] = -t; }
    gx[2 * ND + 1] = 0.0;
}
// (F z) of one contact, engines.py:72-78
template <int ND> __device__ inline void f_rows(double mu, const double *z, double *fz)
{
    double sum = 0.0;
    fz[0] = 0.0;
#pragma unroll
    for (int k = 1; k <= 2 * ND; ++k) { fz[k] = z[2 * ND + 1]; sum += z[k]; }
    fz[2 * ND + 1] = mu * z[0] - sum;
}
// u = (F_c + diag(a))^-1 r   (arrow block, closed form).  ia = 1/a = z/s (the IPM's d), ag = a of the cone row.
template <int ND> __device__ inline void w_apply(double mu, const double *ia, double ag, const double *r, double *u)
{
    double den = ag, num = r[2 * ND + 1] - mu * r[0] * ia[0];
#pragma unroll
    for (int k = 1; k <= 2 * ND; ++k) { den += ia[k]; num += r[k] * ia[k]; }
    const double ug = num / den;
    u[0] = r[0] * ia[0];
#pragma unroll
    for (int k = 1; k <= 2 * ND; ++k) u[k] = (r[k] - ug) * ia[k];
    u[2 * ND + 1] = ug;
}
// w = sum_rows sign_r u_r D_r  ( G_c^T u = P_c w )
template <int ND> __device__ inline void w_vec(const Geo<ND> &g, const double *u, double *w)
{
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double acc = u[0] * g.D[0][j];
#pragma unroll
        for (int k = 1; k <= ND; ++k) acc += (u[k] - u[ND + k]) * g.D[k][j];
        w[j] = acc;
    }
}
// C_c with G_c^T (F_c + diag(a))^-1 G_c = P_c C_c P_c^T   (ia = 1/a, ag = a of the cone row)
template <int ND> __device__ inline void c_mat(const Geo<ND> &g, const double *ia, double ag, double *C)
{
    double den = ag, bv[3] = {0, 0, 0};
    const double ian = ia[0];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = ian * g.D[0][i] * g.D[0][j];
#pragma unroll
    for (int k = 1; k <= ND; ++k) {
        const double i1 = ia[k], i2 = ia[ND + k];
        den += i1 + i2;
        const double al = i1 + i2, be = i1 - i2;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            bv[i] += be * g.D[k][i];
#pragma unroll
            for (int j = 0; j < 3; ++j) C[3 * i + j] += al * g.D[k][i] * g.D[k][j];
        }
    }
    const double iden = 1.0 / den;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] -= bv[i] * (bv[j] - g.mu * ian * g.D[0][j]) * iden;
}

// ---- LDS-resident state of one scene ------------------------------------------------------
struct Lds {
    double *K;      // n x lda
    double *pbuf;   // [maxc][6]  p1, p2
    double *cw;     // [maxc][9]  C matrices  /  [maxc][6] two 3-vector payloads
    double *xv;     // [n] current (x, y)
    double *g1, *g2;// [nz] gathered G^T products
    double *sol;    // [n] latest solve result
    double *dxa;    // [n] affine direction
    double *pl;     // [nz] linear term
    int *piv;       // [n]
    int *cb;        // [2][maxc] body ids (LDS copy)
    int *chunk;     // [maxc] first contact of every piece of at most chunk_len(nc) consecutive contacts of one (body1, body2) run
    int nchunks;
    int n, kn, lda, nz, neq, nb, maxc, nc;
    double *kf;     // global: [n][64] factored rows parked between the solves of an iteration (register path)
    const double *Ag;   // global equality rows [neq][nz]
};

// contacts per piece of a run (assemble_K: four lanes per piece, one per 3-row band; the gathers: one lane per piece).  With many
// contacts longer pieces keep the number of lanes that meet at an LDS address down and the 4 x pieces tasks of the assembly within
// one round of the wavefront (84 contacts in ~14 runs: 0.543 / 0.537 / 0.533 ms per launch for 4 / 6 / 8); with few contacts every
// lane's chain should be as short as possible (config 2: <= 8 contacts)
#if !defined(DSS_CHUNK_BIG)
#define DSS_CHUNK_BIG 8
#endif
__device__ inline int chunk_len(int nc) { return nc > 48 ? DSS_CHUNK_BIG : (nc > 24 ? 2 : 1); }
// sizes with a register-resident factor/solve: only H = Q + sum P C P^T is assembled in LDS, the equality rows join
// in registers and the factored rows are parked in the (L2-resident) workspace between the two solves of an iteration
__host__ __device__ inline bool reg_path(int n) { return n == 54 || n == 18; }
__host__ __device__ inline size_t lds_doubles(int nb, int neq, int maxc)
{
    const int nz = 6 * nb, n = nz + neq, kn = reg_path(n) ? nz : n, lda = kn | 1;
    return (size_t)kn * lda + 6 * (size_t)maxc + 9 * (size_t)maxc + 3 * (size_t)n + 3 * (size_t)nz + 4;
}
__host__ __device__ inline size_t lds_bytes(int nb, int neq, int maxc)
{
    const int n = 6 * nb + neq;
    return lds_doubles(nb, neq, maxc) * 8 + (size_t)(n + 3 * maxc + 4) * 4;
}
__device__ inline void carve_lds(Lds &L, double *base, int nb, int neq, int maxc)
{
    L.nb = nb; L.neq = neq; L.maxc = maxc; L.nz = 6 * nb; L.n = L.nz + neq;
    L.kn = reg_path(L.n) ? L.nz : L.n;
    L.lda = L.kn | 1;
    double *q = base;
    L.K = q; q += (size_t)L.kn * L.lda;
    L.pbuf = q; q += 6 * maxc;
    L.cw = q; q += 9 * maxc;
    L.xv = q; q += L.n;
    L.sol = q; q += L.n;
    L.dxa = q; q += L.n;
    L.g1 = q; q += L.nz;
    L.g2 = q; q += L.nz;
    L.pl = q; q += L.nz;
    q += 4;
    int *ip = reinterpret_cast<int *>(q);
    L.piv = ip; ip += L.n;
    L.cb = ip; ip += 2 * maxc;
    L.chunk = ip;
    L.nchunks = 0;
}

// Pieces for assemble_K: detection emits contacts pair by pair, so the contacts of one (body1, body2) form a run; a run is cut
// into pieces of at most chunk_len(nc) contacts.  Contact c opens a piece if it opens a run or sits a multiple of that behind the
// start of its run (inclusive prefix maximum of the run starts); ordered compaction of the openers by ballot prefix sums.
// Needs the body ids in L.cb.
__device__ void build_chunks(Lds &L, int nc)
{
    const int lane = lane_id();
    const int *cbody = L.cb;
    const int CHUNK = chunk_len(nc);
    int nch = 0, carry = -1;
    for (int base = 0; base < nc; base += WAVE) {
        const int c = base + lane;
        const bool valid = c < nc;
        const bool opens_run = valid && (c == 0 || cbody[c] != cbody[c - 1] || cbody[L.maxc + c] != cbody[L.maxc + c - 1]);
        int rs = opens_run ? c : -1;
        for (int o = 1; o < WAVE; o <<= 1) {
            const int up = __shfl_up(rs, o, WAVE);
            if (lane >= o && up > rs) rs = up;
        }
        if (carry > rs) rs = carry;
        carry = __shfl(rs, WAVE - 1, WAVE);
        const int opens = valid && ((c - rs) % CHUNK == 0);
        const unsigned long long m = __ballot(opens);
        if (opens) L.chunk[nch + __popcll(m & ((1ull << lane) - 1ull))] = c;
        nch += __popcll(m);
    }
    L.nchunks = nch;
    L.nc = nc;
    __syncthreads();
}

// body ids of the contacts in LDS + the pieces of their runs
__device__ void build_lists(Lds &L, const int *cbody_g, int nc)
{
    const int lane = lane_id();
    for (int c = lane; c < nc; c += WAVE) { L.cb[c] = cbody_g[c]; L.cb[L.maxc + c] = cbody_g[L.maxc + c]; }
    __syncthreads();
    build_chunks(L, nc);
}

// out[6b..6b+5] = sum over contacts of body b of  +-[p x w, w]   (NP payloads of 3 doubles in L.cw)
// Lanes = pieces of runs, as in assemble_K: a lane sums [p1 x w, w] and [p2 x w] over its own contacts and adds the three
// sums to the two bodies of its run with LDS atomics.  (Before: one lane per BODY walked the body's whole contact list, ~24
// dependent steps for a box in the middle of a stack, twice per interior-point iteration.)
template <int NP> __device__ void gather(const Lds &L, double *out0, double *out1)
{
    const int lane = lane_id();
    const int *cbody = L.cb;
    for (int i = lane; i < L.nz; i += WAVE) { out0[i] = 0.0; if (NP > 1) out1[i] = 0.0; }
    __syncthreads();
    for (int base = 0; base < L.nchunks; base += WAVE) {
        const int ci = base + lane;
        const bool on = ci < L.nchunks;
        int c0 = 0, c1 = 0, b1 = 0, b2 = 0;
        if (on) {
            c0 = L.chunk[ci];
            c1 = ci + 1 < L.nchunks ? L.chunk[ci + 1] : L.nc;
            b1 = cbody[c0]; b2 = cbody[L.maxc + c0];
        }
        double sw[NP][3], sx1[NP][3], sx2[NP][3];
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int j = 0; j < 3; ++j) { sw[q][j] = 0.0; sx1[q][j] = 0.0; sx2[q][j] = 0.0; }
        for (int c = c0; c < c1; ++c) {
            const double *pp = L.pbuf + 6 * c;
            const double p1[3] = {pp[0], pp[1], pp[2]}, p2[3] = {pp[3], pp[4], pp[5]};
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const double *wp = L.cw + (size_t)(3 * NP) * c + 3 * q;
                const double w[3] = {wp[0], wp[1], wp[2]};
                double x1[3], x2[3];
                cross3(p1, w, x1);
                cross3(p2, w, x2);
#pragma unroll
                for (int j = 0; j < 3; ++j) { sw[q][j] += w[j]; sx1[q][j] += x1[j]; sx2[q][j] += x2[j]; }
            }
        }
        if (on) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                if (q != 0 && q != NP - 1) continue;
                double *o = q == 0 ? out0 : out1;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    atomicAdd(o + 6 * b1 + j, sx1[q][j]);
                    atomicAdd(o + 6 * b1 + 3 + j, sw[q][j]);
                    atomicAdd(o + 6 * b2 + j, -sx2[q][j]);
                    atomicAdd(o + 6 * b2 + 3 + j, -sw[q][j]);
                }
            }
        }
    }
    __syncthreads();
}

// K = [[Q + sum_c P C P^T, A^T],[A, 0]]  (C matrices in L.cw, 9 per contact)
//
// The contact terms.  P_c = [X(p1); I; -X(p2); -I] (12 x 3), so the 12 x 12 block of a contact is made of
//   rows of  A C  with A in {X(p1), I, X(p2), I}   times   {X(p1)^T, I, X(p2)^T, I}:   (y X(p)^T) = p x y  for a row y.
// Lanes = PIECES of at most chunk_len(nc) consecutive contacts of one (body1, body2) run: a lane forms the terms of its own contacts
// and sums them in registers, one 3-row band of the 12 x 12 block at a time (27 running sums), without a word to any other
// lane; the bands are then added into K with LDS atomics (one wavefront, one instruction stream: lanes that meet at an
// address are served in a fixed order, program order settles the rest -- the sum is reproducible run to run).  Before, 48
// lanes each owned three entries and walked ALL contacts of the scene one after the other: 84 dependent steps of ~13 LDS
// reads each per assembly, a third of an interior-point iteration (26 -> 15 us).
__device__ void assemble_K(Lds &L, const double *Mblk, const double *A, const int * /*cbody_g*/, int nc)
{
    const int *cbody = L.cb;
    const int lane = lane_id(), n = L.n, lda = L.lda, nz = L.nz;
    for (int e = lane; e < L.kn * lda; e += WAVE) L.K[e] = 0.0;
    __syncthreads();
    for (int e = lane; e < L.nb * 36; e += WAVE) {
        const int b = e / 36, i = (e % 36) / 6, j = e % 6;
        L.K[(6 * b + i) * lda + 6 * b + j] = Mblk[e];
    }
    if (L.kn == n)   // LDS fallback path keeps the whole KKT matrix
        for (int e = lane; e < L.neq * nz; e += WAVE) {
            const int i = e / nz, j = e % nz;
            const double v = A[e];
            L.K[(nz + i) * lda + j] = v;
            L.K[j * lda + nz + i] = v;
        }
    __syncthreads();
    // lane = (piece, band): the four 3-row bands of a piece's 12 x 12 block -- rows of body 1 (rotation, translation), of body 2
    // (rotation, translation) -- go to four neighbouring lanes (they read the same C and p: LDS broadcasts), so the chain a
    // lane walks is its piece's contacts once, not four times
    const int ntask = 4 * L.nchunks;
    for (int base = 0; base < ntask; base += WAVE) {
        const int ti = base + lane, ci = ti >> 2, rp = ti & 3;
        const bool on = ti < ntask;
        int c0 = 0, c1 = 0, b1 = 0, b2 = 0;
        if (on) {
            c0 = L.chunk[ci];
            c1 = ci + 1 < L.nchunks ? L.chunk[ci + 1] : nc;
            b1 = cbody[c0]; b2 = cbody[L.maxc + c0];
        }
        const bool rot = (rp & 1) == 0, first = rp < 2;
        double sx1[3][3], sa[3][3], sx2[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) { sx1[i][j] = 0.0; sa[i][j] = 0.0; sx2[i][j] = 0.0; }
        for (int c = c0; c < c1; ++c) {
            const double *Cp = L.cw + 9 * c, *pp = L.pbuf + 6 * c;
            double C[9], p1[3], p2[3], pr[3];
#pragma unroll
            for (int j = 0; j < 9; ++j) C[j] = Cp[j];
#pragma unroll
            for (int j = 0; j < 3; ++j) { p1[j] = pp[j]; p2[j] = pp[3 + j]; pr[j] = first ? p1[j] : p2[j]; }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int ia = (i + 2) % 3, ib = (i + 1) % 3;
                double a[3], x1[3], x2[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {    // row i of X(p) C = (p x C[:, j])[i], or row i of C
                    const double xr = pr[ib] * C[3 * ia + j] - pr[ia] * C[3 * ib + j];
                    a[j] = rot ? xr : C[3 * i + j];
                }
                cross3(p1, a, x1);
                cross3(p2, a, x2);
#pragma unroll
                for (int j = 0; j < 3; ++j) { sx1[i][j] += x1[j]; sa[i][j] += a[j]; sx2[i][j] += x2[j]; }
            }
        }
        if (on) {
            const int rb = first ? b1 : b2;
            const double s1 = first ? 1.0 : -1.0, s2 = -s1;      // sign of the band against body 1's / body 2's columns
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double *row = L.K + (6 * rb + (rot ? 0 : 3) + i) * lda;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    atomicAdd(row + 6 * b1 + j, s1 * sx1[i][j]);
                    atomicAdd(row + 6 * b1 + 3 + j, s1 * sa[i][j]);
                    atomicAdd(row + 6 * b2 + j, s2 * sx2[i][j]);
                    atomicAdd(row + 6 * b2 + 3 + j, s2 * sa[i][j]);
                }
            }
        }
    }
    __syncthreads();
}

// In-place partial-pivot LU of K (n <= 64) in LDS.
__device__ void factor_K(Lds &L)
{
    const int lane = lane_id(), n = L.n, lda = L.lda;
    double *K = L.K;
    for (int k = 0; k < n; ++k) {
        double v = (lane >= k && lane < n) ? fabs(K[lane * lda + k]) : -1.0;
        int p = lane;
        wave_argmax(v, p);
        if (lane == 0) L.piv[k] = p;
        if (p != k && lane < n) {
            const double t = K[k * lda + lane];
            K[k * lda + lane] = K[p * lda + lane];
            K[p * lda + lane] = t;
        }
        __syncthreads();
        const double inv = 1.0 / K[k * lda + k];
        if (lane > k && lane < n) K[lane * lda + k] *= inv;
        __syncthreads();
        const int m = n - k - 1;
        if (m > 0) {
            // 2-D lane map: w lanes across a row, 64/w rows per pass
            const int w = m > 32 ? 64 : (m > 16 ? 32 : (m > 8 ? 16 : 8));
            const int rpp = 64 / w, lr = lane / w, lc = lane % w;
            for (int i0 = 0; i0 < m; i0 += rpp) {
                const int i = k + 1 + i0 + lr, j = k + 1 + lc;
                if (i < n && j < n) K[i * lda + j] -= K[i * lda + k] * K[k * lda + j];
            }
        }
        __syncthreads();
    }
}

// Solve K x = rhs with rhs/x distributed one entry per lane (lane i <-> entry i).
__device__ double solve_K(const Lds &L, double x)
{
    const int lane = lane_id(), n = L.n, lda = L.lda;
    const double *K = L.K;
    for (int k = 0; k < n; ++k) {
        const int p = L.piv[k];
        const double xk = __shfl(x, k, WAVE), xp = __shfl(x, p, WAVE);
        if (p != k) { if (lane == k) x = xp; else if (lane == p) x = xk; }
    }
    for (int k = 0; k < n - 1; ++k) {
        const double xk = __shfl(x, k, WAVE);
        if (lane > k && lane < n) x -= K[lane * lda + k] * xk;
    }
    for (int k = n - 1; k >= 0; --k) {
        if (lane == k) x /= K[k * lda + k];
        const double xk = __shfl(x, k, WAVE);
        if (lane < k) x -= K[lane * lda + k] * xk;
    }
    return x;
}


// K factor + solves: register-resident for the compiled sizes (n = 54: 8 bodies + 6 equality rows, the
// configuration BASELINE.json is quoted on; n = 18: 2 bodies), LDS fallback for any other n <= 64.
template <int N> __device__ double kkt_factor_solve_reg(Lds &L, double rhs)
{
    RegK<N> R;
    const int lane = lane_id(), nz = L.nz;
    // row `lane` of K = [[H, A^T],[A, 0]]: H from LDS, equality rows straight from global
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double v = 0.0;
        if (lane < nz) v = (j < nz) ? L.K[lane * L.lda + j] : L.Ag[(j - nz) * nz + lane];
        else if (lane < N) v = (j < nz) ? L.Ag[(lane - nz) * nz + j] : 0.0;
        R.a[j] = v;
    }
    regk_factor_natural<N>(R);
    const double x = regk_solve_natural<N>(R, rhs);
    if (L.kf) {   // the forward pass solves a second time with the same factors (corrector)
#pragma unroll
        for (int j = 0; j < N; ++j) L.kf[(size_t)j * WAVE + lane] = R.a[j];
    }
    __syncthreads();
    return x;
}
template <int N> __device__ double kkt_solve_reg(Lds &L, double rhs)
{
    RegK<N> R;
    const int lane = lane_id();
#pragma unroll
    for (int j = 0; j < N; ++j) R.a[j] = L.kf[(size_t)j * WAVE + lane];
    return regk_solve_natural<N>(R, rhs);
}
__device__ double kkt_factor_solve(Lds &L, double rhs)
{
    if (L.n == 54) return kkt_factor_solve_reg<54>(L, rhs);
    if (L.n == 18) return kkt_factor_solve_reg<18>(L, rhs);
    factor_K(L);
    return solve_K(L, rhs);
}
__device__ double kkt_solve(Lds &L, double rhs)
{
    if (L.n == 54) return kkt_solve_reg<54>(L, rhs);
    if (L.n == 18) return kkt_solve_reg<18>(L, rhs);
    return solve_K(L, rhs);
}

// (Q v)_i for the block-diagonal Q
__device__ inline double q_times(const double *Mblk, const double *v, int i)
{
    const int b = i / 6, r = i % 6;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) acc += Mblk[36 * b + 6 * r + j] * v[6 * b + j];
    return acc;
}

// reference get_step (batch.py:234-237) pieces accumulated per lane
struct StepAcc {
    double amax = -INFINITY, amin_keep = INFINITY;
    int any_pos = 0;
    __device__ inline void add(double v, double dv)
    {
        const double a = -v / dv;
        amax = fmax(amax, a);
        if (dv > 0.0) any_pos = 1; else amin_keep = fmin(amin_keep, a);
    }
    __device__ inline double finish()
    {
        const double mx = wave_max_dpp(amax), mn = wave_min_dpp(amin_keep);
        const bool anyp = __ballot(any_pos) != 0ull;
        const double repl = mx > 1.0 ? mx : 1.0;
        return anyp ? fmin(mn, repl) : mn;
    }
};

template <int ND>
__global__ void __launch_bounds__(64)
lcp_contact_forward_kernel(const double *Mblk_, const double *pvec_, const double *A_, const double *bvec_,
                           const double *cop_, const int *cbody_, const int *ncs, const int *active, int nb, int neq,
                           int maxc, double eps, int not_improved_lim, int max_iter, double *x_out_, double *lam_,
                           double *slack_, double *nu_, int *iters, int *status, double *ws_)
{
    constexpr int NR = Geo<ND>::NR, NF = Geo<ND>::NF;
    DSS_DYN_LDS(double, ldsmem);
    const int sc = blockIdx.x, lane = lane_id();
    if (active && !active[sc]) return;
    Lds L;
    carve_lds(L, ldsmem, nb, neq, maxc);
    const int nz = L.nz, n = L.n;
    const double *Mblk = Mblk_ + (size_t)sc * nb * 36, *pvec = pvec_ + (size_t)sc * nz;
    const double *A = neq ? A_ + (size_t)sc * neq * nz : nullptr, *bvec = neq ? bvec_ + (size_t)sc * neq : nullptr;
    const double *cop = cop_ + (size_t)sc * NF * maxc;
    const int *cbody = cbody_ + (size_t)sc * 2 * maxc;
    double *x_out = x_out_ + (size_t)sc * nz, *lam = lam_ + (size_t)sc * NR * maxc, *slack = slack_ + (size_t)sc * NR * maxc;
    double *nu = neq ? nu_ + (size_t)sc * neq : nullptr;
    double *ws = ws_ + (size_t)sc * (5 * NR * maxc + 64 * 64);
    L.kf = ws + (size_t)5 * NR * maxc;
    L.Ag = A;
    double *cs = ws, *cz = ws + (size_t)NR * maxc, *crz = ws + (size_t)2 * NR * maxc, *cds = ws + (size_t)3 * NR * maxc,
           *cdz = ws + (size_t)4 * NR * maxc;
    int nc = ncs[sc];
    if (nc > maxc) nc = maxc;
    const int nineq = nc * NR;

    for (int i = lane; i < nz; i += WAVE) L.pl[i] = pvec[i];
    for (int c = lane; c < nc; c += WAVE) {
        const int o = 3 * (1 + ND);
#pragma unroll
        for (int j = 0; j < 6; ++j) L.pbuf[6 * c + j] = cop[(size_t)(o + j) * maxc + c];
    }
    __syncthreads();
    build_lists(L, cbody, nc);

    // ---- initial point: d = 1  (batch.py:85-110) -------------------------------------------
    for (int c = lane; c < nc; c += WAVE) {
        Geo<ND> g;
        load_geo<ND>(g, cop, cbody, maxc, c);
        double a[NR], t[NR], u[NR], w[3], C[9];
#pragma unroll
        for (int r = 0; r < NR; ++r) { a[r] = 1.0; t[r] = 0.0; }
        t[0] = -g.hn;  // rz - rs/d with rz = -h, rs = 0
        w_apply<ND>(g.mu, a, 1.0, t, u);
        w_vec<ND>(g, u, w);
        c_mat<ND>(g, a, 1.0, C);
        // the C matrices go to K first; stash w in the (not yet used) ds scratch
#pragma unroll
        for (int j = 0; j < 9; ++j) L.cw[9 * c + j] = C[j];
#pragma unroll
        for (int j = 0; j < 3; ++j) cds[(size_t)j * maxc + c] = w[j];
    }
    __syncthreads();
    assemble_K(L, Mblk, A, cbody, nc);
    for (int c = lane; c < nc; c += WAVE)
#pragma unroll
        for (int j = 0; j < 3; ++j) L.cw[3 * c + j] = cds[(size_t)j * maxc + c];
    __syncthreads();
    gather<1>(L, L.g1, nullptr);
    {
        double rhs = 0.0;
        if (lane < nz) rhs = -L.pl[lane] - L.g1[lane];
        else if (lane < n) rhs = bvec[lane - nz];
        const double sol = kkt_factor_solve(L, rhs);
        if (lane < n) L.xv[lane] = sol;
    }
    __syncthreads();
    if (nc == 0) {  // no complementarity conditions: the linear solve is the answer (engines.py:40-54)
        if (lane < nz) x_out[lane] = L.xv[lane];
        else if (lane < n) nu[lane - nz] = L.xv[lane];
        if (lane == 0) { iters[sc] = 0; status[sc] = DSS_LCP_OK; }
        return;
    }
    {
        double mins = INFINITY, minz = INFINITY;
        for (int c = lane; c < nc; c += WAVE) {
            Geo<ND> g;
            load_geo<ND>(g, cop, cbody, maxc, c);
            double a[NR], r[NR], u[NR], vr[3];
            rel_vel<ND>(L.xv, g, vr);
            g_rows<ND>(g, vr, r);
            r[0] -= g.hn;
#pragma unroll
            for (int q = 0; q < NR; ++q) a[q] = 1.0;
            w_apply<ND>(g.mu, a, 1.0, r, u);
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                cz[(size_t)q * maxc + c] = u[q];
                cs[(size_t)q * maxc + c] = -u[q];
                minz = fmin(minz, u[q]);
                mins = fmin(mins, -u[q]);
            }
        }
        mins = wave_min_dpp(mins);
        minz = wave_min_dpp(minz);
        __syncthreads();
        for (int c = lane; c < nc; c += WAVE)
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                if (mins < 0) cs[(size_t)q * maxc + c] -= mins - 1.0;
                if (minz < 0) cz[(size_t)q * maxc + c] -= minz - 1.0;
            }
        __syncthreads();
    }

    double best = 0.0;
    int have_best = 0, not_improved = 0, it = 0;
    LSTAMP_INIT;
    LSTAMP(0);
    for (it = 0; it < max_iter; ++it) {
        // ---- residuals (batch.py:117-131) and the affine right-hand side ---------------------
        double acc_rz = 0.0, acc_sz = 0.0;
        for (int c = opaque_lane(lane); c < nc; c += WAVE) {
            Geo<ND> g;
            load_geo<ND>(g, cop, cbody, maxc, c);
            double s[NR], z[NR], gx[NR], fz[NR], a[NR], t[NR], u[NR], vr[3], w[3];
#pragma unroll
            for (int q = 0; q < NR; ++q) { s[q] = cs[(size_t)q * maxc + c]; z[q] = cz[(size_t)q * maxc + c]; }
            rel_vel<ND>(L.xv, g, vr);
            g_rows<ND>(g, vr, gx);
            f_rows<ND>(g.mu, z, fz);
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const double rz = gx[q] + s[q] - (q == 0 ? g.hn : 0.0) - fz[q];
                crz[(size_t)q * maxc + c] = rz;
                acc_rz += rz * rz;
                acc_sz += s[q] * z[q];
                a[q] = z[q] / s[q];   // 1/a = d
                t[q] = rz - s[q];  // rz - rs/d with rs = z
            }
            w_vec<ND>(g, z, w);
#pragma unroll
            for (int j = 0; j < 3; ++j) L.cw[6 * c + j] = w[j];
            w_apply<ND>(g.mu, a, s[NR - 1] / z[NR - 1], t, u);
            w_vec<ND>(g, u, w);
#pragma unroll
            for (int j = 0; j < 3; ++j) L.cw[6 * c + 3 + j] = w[j];
        }
        acc_rz = wave_sum(acc_rz);
        const double sz = wave_sum(acc_sz);
        __syncthreads();
        LSTAMP(1);
        gather<2>(L, L.g1, L.g2);  // g1 = G^T z, g2 = G^T W (rz - s)
        LSTAMP(2);
        double rx = 0.0, ry = 0.0;
        if (lane < nz) {
            rx = q_times(Mblk, L.xv, lane) + L.pl[lane] + L.g1[lane];
            for (int e = 0; e < neq; ++e) rx += A[e * nz + lane] * L.xv[nz + e];
        } else if (lane < n) {
            const int e = lane - nz;
            for (int j = 0; j < nz; ++j) ry += A[e * nz + j] * L.xv[j];
            ry -= bvec[e];
        }
        const double nrx = sqrt(wave_sum(rx * rx)), nry = sqrt(wave_sum(ry * ry));
        const double mu = fabs(sz / nineq);
        const double resid = sqrt(acc_rz) + nry + nrx + nineq * mu;
        if (!have_best || resid < best) {
            best = resid; have_best = 1; not_improved = 0;
            if (lane < nz) x_out[lane] = L.xv[lane];
            else if (lane < n) nu[lane - nz] = L.xv[lane];
            for (int c = opaque_lane(lane); c < nc; c += WAVE)
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    lam[(size_t)q * maxc + c] = cz[(size_t)q * maxc + c];
                    slack[(size_t)q * maxc + c] = cs[(size_t)q * maxc + c];
                }
        } else {
            ++not_improved;
        }
        if (not_improved == not_improved_lim || best < eps || mu > 1e32) break;
        LSTAMP(3);

        // ---- K(d) and the affine direction (batch.py:135,174) --------------------------------
        __syncthreads();
        for (int c = opaque_lane(lane); c < nc; c += WAVE) {
            Geo<ND> g;
            load_geo<ND>(g, cop, cbody, maxc, c);
            double a[NR], C[9];
#pragma unroll
            for (int q = 0; q < NR; ++q) a[q] = cz[(size_t)q * maxc + c] / cs[(size_t)q * maxc + c];
            c_mat<ND>(g, a, cs[(size_t)(NR - 1) * maxc + c] / cz[(size_t)(NR - 1) * maxc + c], C);
#pragma unroll
            for (int j = 0; j < 9; ++j) L.cw[9 * c + j] = C[j];
        }
        __syncthreads();
        LSTAMP(4);
        assemble_K(L, Mblk, A, cbody, nc);
        LSTAMP(5);
        LSTAMP(6);
        {
            double rhs = 0.0;
            if (lane < nz) rhs = -rx - L.g2[lane];
            else if (lane < n) rhs = -ry;
            const double sol = kkt_factor_solve(L, rhs);
            if (lane < n) L.dxa[lane] = sol;
        }
        __syncthreads();
        LSTAMP(7);
        StepAcc stz, sts;
        for (int c = opaque_lane(lane); c < nc; c += WAVE) {
            Geo<ND> g;
            load_geo<ND>(g, cop, cbody, maxc, c);
            double s[NR], z[NR], a[NR], r[NR], u[NR], vr[3];
            rel_vel<ND>(L.dxa, g, vr);
            g_rows<ND>(g, vr, r);
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                s[q] = cs[(size_t)q * maxc + c]; z[q] = cz[(size_t)q * maxc + c];
                a[q] = z[q] / s[q];
                r[q] += crz[(size_t)q * maxc + c] - s[q];
            }
            w_apply<ND>(g.mu, a, s[NR - 1] / z[NR - 1], r, u);
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const double dz = u[q], ds = (-z[q] - dz) / a[q];
                cdz[(size_t)q * maxc + c] = dz;
                cds[(size_t)q * maxc + c] = ds;
                stz.add(z[q], dz);
                sts.add(s[q], ds);
            }
        }
        double alpha = fmin(fmin(stz.finish(), sts.finish()), 1.0);
        __syncthreads();
        LSTAMP(8);
        double t3 = 0.0;
        for (int c = opaque_lane(lane); c < nc; c += WAVE)
#pragma unroll
            for (int q = 0; q < NR; ++q)
                t3 += (cs[(size_t)q * maxc + c] + alpha * cds[(size_t)q * maxc + c]) *
                      (cz[(size_t)q * maxc + c] + alpha * cdz[(size_t)q * maxc + c]);
        t3 = wave_sum(t3);
        double sig = t3 / sz;
        sig = sig * sig * sig;
        // ---- corrector (batch.py:194-205): rx = rz = ry = 0, rs = (-mu sig + ds dz)/s ----------
        for (int c = opaque_lane(lane); c < nc; c += WAVE) {
            Geo<ND> g;
            load_geo<ND>(g, cop, cbody, maxc, c);
            double a[NR], t[NR], u[NR], w[3], ag = 1.0;
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const double s = cs[(size_t)q * maxc + c], z = cz[(size_t)q * maxc + c];
                const double rs2 = (-mu * sig + cds[(size_t)q * maxc + c] * cdz[(size_t)q * maxc + c]) / s;
                a[q] = z / s;
                t[q] = rs2 / a[q];
                if (q == NR - 1) ag = s / z;
            }
            w_apply<ND>(g.mu, a, ag, t, u);
            w_vec<ND>(g, u, w);
#pragma unroll
            for (int j = 0; j < 3; ++j) L.cw[3 * c + j] = w[j];
        }
        __syncthreads();
        LSTAMP(9);
        gather<1>(L, L.g1, nullptr);
        {
            double rhs = (lane < nz) ? L.g1[lane] : 0.0;
            const double sol = kkt_solve(L, rhs);
            if (lane < n) L.sol[lane] = sol;
        }
        __syncthreads();
        LSTAMP(10);
        StepAcc stz2, sts2;
        for (int c = opaque_lane(lane); c < nc; c += WAVE) {
            Geo<ND> g;
            load_geo<ND>(g, cop, cbody, maxc, c);
            double s[NR], z[NR], a[NR], r[NR], u[NR], rs2[NR], vr[3];
            rel_vel<ND>(L.sol, g, vr);
            g_rows<ND>(g, vr, r);
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                s[q] = cs[(size_t)q * maxc + c]; z[q] = cz[(size_t)q * maxc + c];
                rs2[q] = (-mu * sig + cds[(size_t)q * maxc + c] * cdz[(size_t)q * maxc + c]) / s[q];
                a[q] = z[q] / s[q];
                r[q] -= rs2[q] / a[q];
            }
            w_apply<ND>(g.mu, a, s[NR - 1] / z[NR - 1], r, u);
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const double dz = cdz[(size_t)q * maxc + c] + u[q];
                const double ds = cds[(size_t)q * maxc + c] + (-rs2[q] - u[q]) / a[q];
                cdz[(size_t)q * maxc + c] = dz;
                cds[(size_t)q * maxc + c] = ds;
                stz2.add(z[q], dz);
                sts2.add(s[q], ds);
            }
        }
        alpha = fmin(0.999 * fmin(stz2.finish(), sts2.finish()), 1.0);
        __syncthreads();
        LSTAMP(11);
        if (lane < n) L.xv[lane] += alpha * (L.dxa[lane] + L.sol[lane]);
        for (int c = opaque_lane(lane); c < nc; c += WAVE)
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                cs[(size_t)q * maxc + c] += alpha * cds[(size_t)q * maxc + c];
                cz[(size_t)q * maxc + c] += alpha * cdz[(size_t)q * maxc + c];
            }
        __syncthreads();
        LSTAMP(12);
    }
    if (lane == 0) { iters[sc] = it; status[sc] = (best > 1.0) ? DSS_LCP_INACCURATE : DSS_LCP_OK; }
}

// The same solver with the per-contact IPM state (s, z, rz, ds, dz, d = z/s) held in registers: lane l owns contacts
// l and l + 64 (maxc <= 128), every loop over them is unrolled with compile-time indices.  Nothing but the contact
// operands is re-read from memory inside the iteration, d is divided out once per iteration instead of once per
// pass, and sum(s z) after the affine step needs no pass at all.  Arithmetic per contact is expression for
// expression that of lcp_contact_forward_kernel (the streaming form, kept for maxc > 128): results are bit-identical.
// N = compiled size of the register-resident factorisation (54, 18) or 0 = LDS fallback for other n <= 64.  With
// N > 0 the factored rows stay in registers from the affine solve to the corrector solve of the same iteration.
template <int ND, int N>
__global__ void __launch_bounds__(64)
lcp_contact_forward_reg_kernel(const double *Mblk_, const double *pvec_, const double *A_, const double *bvec_,
                               const double *cop_, const int *cbody_, const int *ncs, const int *active, int nb, int neq,
                               int maxc, double eps, int not_improved_lim, int max_iter, double *x_out_, double *lam_,
                               double *slack_, double *nu_, int *iters, int *status, double *ws_)
{
    constexpr int NR = Geo<ND>::NR, NF = Geo<ND>::NF, CPL = 2;
    DSS_DYN_LDS(double, ldsmem);
    const int sc = blockIdx.x, lane = lane_id();
    if (active && !active[sc]) return;
    Lds L;
    carve_lds(L, ldsmem, nb, neq, maxc);
    const int nz = L.nz, n = L.n;
    const double *Mblk = Mblk_ + (size_t)sc * nb * 36, *pvec = pvec_ + (size_t)sc * nz;
    const double *A = neq ? A_ + (size_t)sc * neq * nz : nullptr, *bvec = neq ? bvec_ + (size_t)sc * neq : nullptr;
    const double *cop = cop_ + (size_t)sc * NF * maxc;
    const int *cbody = cbody_ + (size_t)sc * 2 * maxc;
    double *x_out = x_out_ + (size_t)sc * nz, *lam = lam_ + (size_t)sc * NR * maxc, *slack = slack_ + (size_t)sc * NR * maxc;
    double *nu = neq ? nu_ + (size_t)sc * neq : nullptr;
    L.kf = N > 0 ? nullptr : ws_ + (size_t)sc * (5 * NR * maxc + 64 * 64) + (size_t)5 * NR * maxc;
    L.Ag = A;
    int nc = ncs[sc];
    if (nc > maxc) nc = maxc;
    const int nineq = nc * NR;

    double s[CPL][NR], z[CPL][NR], rzv[CPL][NR], dsv[CPL][NR], dzv[CPL][NR], av[CPL][NR];
    bool valid[CPL];
#pragma unroll
    for (int r = 0; r < CPL; ++r) valid[r] = lane + WAVE * r < nc;

    for (int i = lane; i < nz; i += WAVE) L.pl[i] = pvec[i];
    for (int c = lane; c < nc; c += WAVE) {
        const int o = 3 * (1 + ND);
#pragma unroll
        for (int j = 0; j < 6; ++j) L.pbuf[6 * c + j] = cop[(size_t)(o + j) * maxc + c];
    }
    __syncthreads();
    build_lists(L, cbody, nc);

    // ---- initial point: d = 1  (batch.py:85-110) -------------------------------------------
    {
        double w0[CPL][3];
#pragma unroll
        for (int r = 0; r < CPL; ++r) {
            const int c = lane + WAVE * r;
            if (!valid[r]) continue;
            Geo<ND> g;
            load_geo<ND>(g, cop, cbody, maxc, c);
            double a[NR], t[NR], u[NR], C[9];
#pragma unroll
            for (int q = 0; q < NR; ++q) { a[q] = 1.0; t[q] = 0.0; }
            t[0] = -g.hn;  // rz - rs/d with rz = -h, rs = 0
            w_apply<ND>(g.mu, a, 1.0, t, u);
            w_vec<ND>(g, u, w0[r]);
            c_mat<ND>(g, a, 1.0, C);
#pragma unroll
            for (int j = 0; j < 9; ++j) L.cw[9 * c + j] = C[j];
        }
        __syncthreads();
        assemble_K(L, Mblk, A, cbody, nc);
#pragma unroll
        for (int r = 0; r < CPL; ++r)
            if (valid[r])
#pragma unroll
                for (int j = 0; j < 3; ++j) L.cw[3 * (lane + WAVE * r) + j] = w0[r][j];
        __syncthreads();
    }
    gather<1>(L, L.g1, nullptr);
    {
        double rhs = 0.0;
        if (lane < nz) rhs = -L.pl[lane] - L.g1[lane];
        else if (lane < n) rhs = bvec[lane - nz];
        const double sol = kkt_factor_solve(L, rhs);
        if (lane < n) L.xv[lane] = sol;
    }
    __syncthreads();
    if (nc == 0) {  // no complementarity conditions: the linear solve is the answer (engines.py:40-54)
        if (lane < nz) x_out[lane] = L.xv[lane];
        else if (lane < n) nu[lane - nz] = L.xv[lane];
        if (lane == 0) { iters[sc] = 0; status[sc] = DSS_LCP_OK; }
        return;
    }
    {
        double mins = INFINITY, minz = INFINITY;
#pragma unroll
        for (int r = 0; r < CPL; ++r) {
            if (!valid[r]) continue;
            Geo<ND> g;
            load_geo<ND>(g, cop, cbody, maxc, lane + WAVE * r);
            double a[NR], rr[NR], u[NR], vr[3];
            rel_vel<ND>(L.xv, g, vr);
            g_rows<ND>(g, vr, rr);
            rr[0] -= g.hn;
#pragma unroll
            for (int q = 0; q < NR; ++q) a[q] = 1.0;
            w_apply<ND>(g.mu, a, 1.0, rr, u);
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                z[r][q] = u[q];
                s[r][q] = -u[q];
                minz = fmin(minz, u[q]);
                mins = fmin(mins, -u[q]);
            }
        }
        mins = wave_min_dpp(mins);
        minz = wave_min_dpp(minz);
#pragma unroll
        for (int r = 0; r < CPL; ++r)
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                if (mins < 0) s[r][q] -= mins - 1.0;
                if (minz < 0) z[r][q] -= minz - 1.0;
            }
    }

    // A body pinned by identity equality rows (TotalConstraint3D on body 0: A = [I6 | 0], the floor of every scene of
    // the reference) decouples: x_0 = rhs_y, the other bodies solve H_oo x_o = rhs_o - H_o0 x_0 on their own, and the
    // multipliers follow from the pinned body's rows, y = rhs_0 - H_00 x_0 - H_0o x_o.  Only the (N - 12)-square H_oo
    // is factored then -- 40 % of the elimination work of the full system.
    bool pinned0 = false;
    if constexpr (N > 12) {
        bool okl = true;
        if (lane < nz)
            for (int e = 0; e < 6; ++e) okl = okl && (A[e * nz + lane] == (lane == e ? 1.0 : 0.0));
        pinned0 = neq == 6 && __ballot(!okl) == 0ull;
    }

    // every contact between neighbours in the body order, or with the pinned body: H_oo is block tridiagonal (kkt_reg.h)
    bool tri = false;
    if constexpr (N > 12) {
        bool okc = true;
        for (int c = lane; c < nc; c += WAVE) {
            const int b1 = L.cb[c], b2 = L.cb[maxc + c];
            okc = okc && (b1 == 0 || b2 == 0 || b1 - b2 == 1 || b2 - b1 == 1);
        }
        tri = pinned0 && __ballot(!okc) == 0ull;
        // (the decoupled form below is built for that shape only; a pinned scene whose contacts skip a body takes the general
        //  elimination of the whole system like an unpinned one)
        pinned0 = tri;
    }

    double best = 0.0;
    int have_best = 0, not_improved = 0, it = 0;
    LSTAMP_INIT;
    LSTAMP(0);
    for (it = 0; it < max_iter; ++it) {
        // ---- residuals (batch.py:117-131), the affine right-hand side and K(d) ------------------
        double acc_rz = 0.0, acc_sz = 0.0;
        const int l0 = opaque_lane(lane);
#pragma unroll
        for (int r = 0; r < CPL; ++r) {
            const int c = l0 + WAVE * r;
            if (!valid[r]) continue;
            Geo<ND> g;
            load_geo<ND>(g, cop, cbody, maxc, c);
            double gx[NR], fz[NR], t[NR], u[NR], vr[3], w[3];
            rel_vel<ND>(L.xv, g, vr);
            g_rows<ND>(g, vr, gx);
            f_rows<ND>(g.mu, z[r], fz);
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const double rz = gx[q] + s[r][q] - (q == 0 ? g.hn : 0.0) - fz[q];
                rzv[r][q] = rz;
                acc_rz += rz * rz;
                acc_sz += s[r][q] * z[r][q];
                av[r][q] = z[r][q] / s[r][q];   // 1/a = d
                t[q] = rz - s[r][q];            // rz - rs/d with rs = z
            }
            w_vec<ND>(g, z[r], w);
#pragma unroll
            for (int j = 0; j < 3; ++j) L.cw[6 * c + j] = w[j];
            const double ag = s[r][NR - 1] / z[r][NR - 1];
            w_apply<ND>(g.mu, av[r], ag, t, u);
            w_vec<ND>(g, u, w);
#pragma unroll
            for (int j = 0; j < 3; ++j) L.cw[6 * c + 3 + j] = w[j];
        }
        acc_rz = wave_sum(acc_rz);
        const double sz = wave_sum(acc_sz);
        __syncthreads();
        LSTAMP(1);
        gather<2>(L, L.g1, L.g2);  // g1 = G^T z, g2 = G^T W (rz - s)
        LSTAMP(2);
        double rx = 0.0, ry = 0.0;
        if (N > 12 && pinned0) {      // A = [I6 | 0]: A^T y and A x are copies (the same values as the sums below, whose other terms are exact zeros)
            if (lane < nz) rx = q_times(Mblk, L.xv, lane) + L.pl[lane] + L.g1[lane] + (lane < 6 ? L.xv[nz + lane] : 0.0);
            else if (lane < n) ry = L.xv[lane - nz] - bvec[lane - nz];
        } else if (lane < nz) {
            rx = q_times(Mblk, L.xv, lane) + L.pl[lane] + L.g1[lane];
            for (int e = 0; e < neq; ++e) rx += A[e * nz + lane] * L.xv[nz + e];
        } else if (lane < n) {
            const int e = lane - nz;
            for (int j = 0; j < nz; ++j) ry += A[e * nz + j] * L.xv[j];
            ry -= bvec[e];
        }
        const double nrx = sqrt(wave_sum(rx * rx)), nry = sqrt(wave_sum(ry * ry));
        const double mu = fabs(sz / nineq);
        const double resid = sqrt(acc_rz) + nry + nrx + nineq * mu;
        if (!have_best || resid < best) {
            best = resid; have_best = 1; not_improved = 0;
            if (lane < nz) x_out[lane] = L.xv[lane];
            else if (lane < n) nu[lane - nz] = L.xv[lane];
#pragma unroll
            for (int r = 0; r < CPL; ++r)
                if (valid[r])
#pragma unroll
                    for (int q = 0; q < NR; ++q) {
                        lam[(size_t)q * maxc + l0 + WAVE * r] = z[r][q];
                        slack[(size_t)q * maxc + l0 + WAVE * r] = s[r][q];
                    }
        } else {
            ++not_improved;
        }
        if (not_improved == not_improved_lim || best < eps || mu > 1e32) break;
        LSTAMP(3);

        // ---- K(d) and the affine direction (batch.py:135,174) --------------------------------
        __syncthreads();
        // the contacts' C matrices, formed here rather than carried in registers (18 doubles per lane) through the gathers
        // and the residual block above: the operands come from L2 again, the registers are what this kernel is short of
#pragma unroll
        for (int r = 0; r < CPL; ++r)
            if (valid[r]) {
                Geo<ND> g;
                load_geo<ND>(g, cop, cbody, maxc, l0 + WAVE * r);
                double C[9];
                c_mat<ND>(g, av[r], s[r][NR - 1] / z[r][NR - 1], C);
#pragma unroll
                for (int j = 0; j < 9; ++j) L.cw[9 * (l0 + WAVE * r) + j] = C[j];
            }
        __syncthreads();
        LSTAMP(4);
        assemble_K(L, Mblk, A, cbody, nc);
        LSTAMP(5);
        RegK<(N > 0 ? N : 1)> R;
        {
            double rhs = 0.0, sol;
            if (lane < nz) rhs = -rx - L.g2[lane];
            else if (lane < n) rhs = -ry;
            if (N > 12 && pinned0) {
                constexpr int M = N > 12 ? N - 12 : 1;
                if (lane < n) L.sol[lane] = rhs;     // staging: the right-hand side by index
                __syncthreads();
#pragma unroll
                for (int j = 0; j < M; ++j) R.a[j] = lane < M ? L.K[(lane + 6) * L.lda + j + 6] : 0.0;
                double ro = 0.0;
                if (lane < M) {       // rhs_o - H_o0 x_0 with x_0 = rhs_y
                    ro = L.sol[lane + 6];
                    for (int i = 0; i < 6; ++i) ro -= L.K[(lane + 6) * L.lda + i] * L.sol[nz + i];
                }
                regk_factor_lead_tri<(N > 0 ? N : 1), M>(R);
                const double xo = regk_solve_lead<(N > 0 ? N : 1), M>(R, ro);
                if (lane < 6) L.dxa[lane] = L.sol[nz + lane];
                if (lane < M) L.dxa[lane + 6] = xo;
                __syncthreads();
                if (lane < 6) {
                    double y = L.sol[lane];
                    for (int i = 0; i < nz; ++i) y -= L.K[lane * L.lda + i] * L.dxa[i];
                    L.dxa[nz + lane] = y;
                }
                __syncthreads();
                sol = lane < n ? L.dxa[lane] : 0.0;
            } else if constexpr (N > 0) {
                // row `lane` of K = [[H, A^T],[A, 0]]: H from LDS, equality rows straight from global
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    double v = 0.0;
                    if (lane < nz) v = (j < nz) ? L.K[lane * L.lda + j] : A[(j - nz) * nz + lane];
                    else if (lane < N) v = (j < nz) ? A[(lane - nz) * nz + j] : 0.0;
                    R.a[j] = v;
                }
                regk_factor_natural<N>(R);
                sol = regk_solve_natural<N>(R, rhs);
                __syncthreads();
            } else {
                sol = kkt_factor_solve(L, rhs);
            }
            if (lane < n) L.dxa[lane] = sol;
        }
        __syncthreads();
        LSTAMP(7);
        StepAcc stz, sts;
        const int l1 = opaque_lane(lane);
#pragma unroll
        for (int r = 0; r < CPL; ++r) {
            if (!valid[r]) continue;
            Geo<ND> g;
            load_geo<ND>(g, cop, cbody, maxc, l1 + WAVE * r);
            double rr[NR], u[NR], vr[3];
            rel_vel<ND>(L.dxa, g, vr);
            g_rows<ND>(g, vr, rr);
#pragma unroll
            for (int q = 0; q < NR; ++q) rr[q] += rzv[r][q] - s[r][q];
            w_apply<ND>(g.mu, av[r], s[r][NR - 1] / z[r][NR - 1], rr, u);
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const double dz = u[q], ds = (-z[r][q] - dz) / av[r][q];
                dzv[r][q] = dz;
                dsv[r][q] = ds;
                stz.add(z[r][q], dz);
                sts.add(s[r][q], ds);
            }
        }
        double alpha = fmin(fmin(stz.finish(), sts.finish()), 1.0);
        LSTAMP(8);
        double t3 = 0.0;
#pragma unroll
        for (int r = 0; r < CPL; ++r)
            if (valid[r])
#pragma unroll
                for (int q = 0; q < NR; ++q) t3 += (s[r][q] + alpha * dsv[r][q]) * (z[r][q] + alpha * dzv[r][q]);
        t3 = wave_sum(t3);
        double sig = t3 / sz;
        sig = sig * sig * sig;
        // ---- corrector (batch.py:194-205): rx = rz = ry = 0, rs = (-mu sig + ds dz)/s ----------
        const int l2 = opaque_lane(lane);
#pragma unroll
        for (int r = 0; r < CPL; ++r) {
            const int c = l2 + WAVE * r;
            if (!valid[r]) continue;
            Geo<ND> g;
            load_geo<ND>(g, cop, cbody, maxc, c);
            double t[NR], u[NR], w[3];
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const double rs2 = (-mu * sig + dsv[r][q] * dzv[r][q]) / s[r][q];
                rzv[r][q] = rs2;                 // rz is not needed any more this iteration
                t[q] = rs2 / av[r][q];
            }
            w_apply<ND>(g.mu, av[r], s[r][NR - 1] / z[r][NR - 1], t, u);
            w_vec<ND>(g, u, w);
#pragma unroll
            for (int j = 0; j < 3; ++j) L.cw[3 * c + j] = w[j];
        }
        __syncthreads();
        LSTAMP(9);
        gather<1>(L, L.g1, nullptr);
        {
            double rhs = (lane < nz) ? L.g1[lane] : 0.0, sol;
            if (N > 12 && pinned0) {      // x_0 = 0, H_oo x_o = g1_o, y = g1_0 - H_0o x_o
                constexpr int M = N > 12 ? N - 12 : 1;
                const double xo = regk_solve_lead<(N > 0 ? N : 1), M>(R, lane < M ? L.g1[lane + 6] : 0.0);
                __syncthreads();
                if (lane < 6) L.sol[lane] = 0.0;
                if (lane < M) L.sol[lane + 6] = xo;
                __syncthreads();
                if (lane < 6) {
                    double y = L.g1[lane];
                    for (int i = 6; i < nz; ++i) y -= L.K[lane * L.lda + i] * L.sol[i];
                    L.sol[nz + lane] = y;
                }
                __syncthreads();
                sol = lane < n ? L.sol[lane] : 0.0;
            } else if constexpr (N > 0) sol = regk_solve_natural<N>(R, rhs);
            else sol = kkt_solve(L, rhs);
            if (lane < n) L.sol[lane] = sol;
        }
        __syncthreads();
        LSTAMP(10);
        StepAcc stz2, sts2;
        const int l3 = opaque_lane(lane);
#pragma unroll
        for (int r = 0; r < CPL; ++r) {
            if (!valid[r]) continue;
            Geo<ND> g;
            load_geo<ND>(g, cop, cbody, maxc, l3 + WAVE * r);
            double rr[NR], u[NR], vr[3];
            rel_vel<ND>(L.sol, g, vr);
            g_rows<ND>(g, vr, rr);
#pragma unroll
            for (int q = 0; q < NR; ++q) rr[q] -= rzv[r][q] / av[r][q];
            w_apply<ND>(g.mu, av[r], s[r][NR - 1] / z[r][NR - 1], rr, u);
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const double dz = dzv[r][q] + u[q];
                const double ds = dsv[r][q] + (-rzv[r][q] - u[q]) / av[r][q];
                dzv[r][q] = dz;
                dsv[r][q] = ds;
                stz2.add(z[r][q], dz);
                sts2.add(s[r][q], ds);
            }
        }
        alpha = fmin(0.999 * fmin(stz2.finish(), sts2.finish()), 1.0);
        __syncthreads();
        LSTAMP(11);
        if (lane < n) L.xv[lane] += alpha * (L.dxa[lane] + L.sol[lane]);
#pragma unroll
        for (int r = 0; r < CPL; ++r)
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                s[r][q] += alpha * dsv[r][q];
                z[r][q] += alpha * dzv[r][q];
            }
        __syncthreads();
        LSTAMP(12);
    }
    if (lane == 0) { iters[sc] = it; status[sc] = (best > 1.0) ? DSS_LCP_INACCURATE : DSS_LCP_OK; }
}

// Implicit backward (lcp.py:156-213) in the same reduced form; gradients come out already
// contracted onto the contact operands (directions, contact points, mu, h_n), the mass blocks
// and the linear term -- the dense dG / dF of the reference are never formed.
template <int ND>
__global__ void __launch_bounds__(64)
lcp_contact_backward_kernel(const double *Mblk_, const double *A_, const double *cop_, const int *cbody_,
                            const int *ncs, const int *active, int nb, int neq, int maxc, const double *x_, const double *lam_,
                            const double *slack_, const double *nu_, const double *dl_dx_, double *dMblk_,
                            double *dpvec_, double *dcop_, double *dA_, double *db_, int rows, const int *slot, int Btot)
{
    // slot != NULL: lam_ / slack_ are the bases of the stepper's tape ([max_sub][B][NR][maxc]) and scene sc reads the record of
    // sub-step slot[sc] in place (the reverse sweep used to copy 40 KB per scene into [B][NR][maxc] arrays first)
    // rows: which rows of G pass gradient to the contact geometry they are built from -- 1 the normal row (Jc), 2 the
    // friction rows (Jf); World3D's stop_contact_grad / stop_friction_grad build the others from detached geometry
    constexpr int NR = Geo<ND>::NR, NF = Geo<ND>::NF;
    DSS_DYN_LDS(double, ldsmem);
    const int sc = blockIdx.x, lane = lane_id();
    if (active && !active[sc]) return;
    Lds L;
    carve_lds(L, ldsmem, nb, neq, maxc);
    const int nz = L.nz, n = L.n;
    const double *Mblk = Mblk_ + (size_t)sc * nb * 36;
    const double *A = neq ? A_ + (size_t)sc * neq * nz : nullptr;
    L.kf = nullptr;
    L.Ag = A;
    const double *cop = cop_ + (size_t)sc * NF * maxc;
    const int *cbody = cbody_ + (size_t)sc * 2 * maxc;
    const size_t rec = slot ? (size_t)slot[sc] * Btot + sc : (size_t)sc;
    const double *lam = lam_ + rec * NR * maxc, *slack = slack_ + rec * NR * maxc;
    double *dcop = dcop_ + (size_t)sc * NF * maxc;
    int nc = ncs[sc];
    if (nc > maxc) nc = maxc;

    for (int c = lane; c < nc; c += WAVE) {
        const int o = 3 * (1 + ND);
#pragma unroll
        for (int j = 0; j < 6; ++j) L.pbuf[6 * c + j] = cop[(size_t)(o + j) * maxc + c];
    }
    for (int c = lane; c < nc; c += WAVE) { L.cb[c] = cbody[c]; L.cb[maxc + c] = cbody[maxc + c]; }
    if (lane < nz) L.xv[lane] = x_[(size_t)sc * nz + lane];
    else if (lane < n) L.xv[lane] = nu_[(size_t)sc * neq + lane - nz];
    __syncthreads();
    build_chunks(L, nc);
    for (int c = lane; c < nc; c += WAVE) {
        Geo<ND> g;
        load_geo<ND>(g, cop, cbody, maxc, c);
        double a[NR], C[9];
#pragma unroll
        for (int q = 0; q < NR; ++q)  // d = clamp(lam)/clamp(slack), lcp.py:176
            a[q] = fmax(lam[(size_t)q * maxc + c], 1e-8) / fmax(slack[(size_t)q * maxc + c], 1e-8);
        c_mat<ND>(g, a, fmax(slack[(size_t)(NR - 1) * maxc + c], 1e-8) / fmax(lam[(size_t)(NR - 1) * maxc + c], 1e-8), C);
#pragma unroll
        for (int j = 0; j < 9; ++j) L.cw[9 * c + j] = C[j];
    }
    __syncthreads();
    assemble_K(L, Mblk, A, cbody, nc);
    {
        const double rhs = (lane < nz) ? -dl_dx_[(size_t)sc * nz + lane] : 0.0;
        const double sol = kkt_factor_solve(L, rhs);
        if (lane < n) L.sol[lane] = sol;
    }
    __syncthreads();
    const double *dx = L.sol, *zx = L.xv;
    // dQ (block diagonal part), dp, dA, db
    for (int e = lane; e < nb * 36; e += WAVE) {
        const int b = e / 36, i = 6 * b + (e % 36) / 6, j = 6 * b + e % 6;
        dMblk_[(size_t)sc * nb * 36 + e] = 0.5 * (dx[i] * zx[j] + zx[i] * dx[j]);
    }
    if (lane < nz) dpvec_[(size_t)sc * nz + lane] = dx[lane];
    if (dA_)
        for (int e = lane; e < neq * nz; e += WAVE) {
            const int i = e / nz, j = e % nz;
            dA_[(size_t)sc * neq * nz + e] = dx[nz + i] * zx[j] + zx[nz + i] * dx[j];
        }
    if (db_ && lane < neq) db_[(size_t)sc * neq + lane] = -dx[nz + lane];
    for (int c = lane; c < maxc; c += WAVE) {
        if (c >= nc) {
            for (int f = 0; f < NF; ++f) dcop[(size_t)f * maxc + c] = 0.0;
            continue;
        }
        Geo<ND> g;
        load_geo<ND>(g, cop, cbody, maxc, c);
        double a[NR], l[NR], r[NR], dl[NR], vrx[3], vrz[3], wl[3], wdl[3], t1[3], t2[3];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            l[q] = lam[(size_t)q * maxc + c];
            a[q] = fmax(l[q], 1e-8) / fmax(slack[(size_t)q * maxc + c], 1e-8);
        }
        rel_vel<ND>(dx, g, vrx);
        rel_vel<ND>(zx, g, vrz);
        g_rows<ND>(g, vrx, r);
        w_apply<ND>(g.mu, a, fmax(slack[(size_t)(NR - 1) * maxc + c], 1e-8) / fmax(l[NR - 1], 1e-8), r, dl);  // dlam = W G dx
        // directions
        const double mn = (rows & 1) ? 1.0 : 0.0, mf = (rows & 2) ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < 3; ++j) dcop[(size_t)j * maxc + c] = mn * (dl[0] * vrz[j] + l[0] * vrx[j]);
#pragma unroll
        for (int k = 1; k <= ND; ++k)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                dcop[(size_t)(3 * k + j) * maxc + c] = mf * ((dl[k] - dl[ND + k]) * vrz[j] + (l[k] - l[ND + k]) * vrx[j]);
        // contact points (through the rows that carry gradient)
        const double dmu = dl[NR - 1] * l[0], dhn = -dl[0];
        if (rows != 3) {
            l[0] *= mn; dl[0] *= mn;
#pragma unroll
            for (int k = 1; k <= 2 * ND; ++k) { l[k] *= mf; dl[k] *= mf; }
        }
        w_vec<ND>(g, l, wl);
        w_vec<ND>(g, dl, wdl);
        const int o = 3 * (1 + ND);
        cross3(wdl, zx + 6 * g.b1, t1);
        cross3(wl, dx + 6 * g.b1, t2);
#pragma unroll
        for (int j = 0; j < 3; ++j) dcop[(size_t)(o + j) * maxc + c] = t1[j] + t2[j];
        cross3(wdl, zx + 6 * g.b2, t1);
        cross3(wl, dx + 6 * g.b2, t2);
#pragma unroll
        for (int j = 0; j < 3; ++j) dcop[(size_t)(o + 3 + j) * maxc + c] = -(t1[j] + t2[j]);
        dcop[(size_t)(o + 6) * maxc + c] = dmu;   // dF[cone_c, normal_c] = dlam_cone lam_n
        dcop[(size_t)(o + 7) * maxc + c] = dhn;   // dh_n = -dlam_n
    }
}

}  // namespace
namespace dss {
int lcp_contact_backward_rows(const double *Mblk, const double *A, const double *cop, const int *cbody, const int *nc,
                              const int *active, int B, int nb, int neq, int maxc, int fric_dirs, const double *x,
                              const double *lam, const double *slack, const double *nu, const double *dl_dx, double *dMblk,
                              double *dpvec, double *dcop, double *dA, double *db, int rows, const int *slot, void *stream);
}
namespace {
inline bool dims_ok(int B, int nb, int neq, int maxc, int fd)
{
    return B > 0 && nb > 0 && neq >= 0 && maxc > 0 && (fd == 4 || fd == 8);
}

}  // namespace

#if defined(DSS_DIAG)
__global__ void set_lcp_stamps_kernel(long long *p) { g_lcp_stamps = p; }
extern "C" void dss_diag_set_lcp_stamps(long long *p, void *stream)
{
    hipLaunchKernelGGL(set_lcp_stamps_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, p);
}
#endif

extern "C" {

size_t dss_lcp_contact_workspace_bytes(int B, int nb, int neq, int maxc, int fric_dirs)
{
    if (!dims_ok(B, nb, neq, maxc, fric_dirs)) return 0;
    return (size_t)B * (5 * (fric_dirs + 2) * maxc + 64 * 64) * sizeof(double);
}

int dss_lcp_contact_forward(const double *Mblk, const double *pvec, const double *A, const double *bvec,
                            const double *cop, const int *cbody, const int *nc, const int *active, int B, int nb,
                            int neq, int maxc, int fric_dirs, double eps, int not_improved_lim, int max_iter, double *x, double *lam,
                            double *slack, double *nu, int *iters, int *status, void *workspace,
                            size_t workspace_bytes, void *stream)
{
    if (!dims_ok(B, nb, neq, maxc, fric_dirs)) return DSS_E_BADARG;
    if (!Mblk || !pvec || !cop || !cbody || !nc || !x || !lam || !slack || !iters || !status || !workspace) return DSS_E_BADARG;
    if (neq > 0 && (!A || !bvec || !nu)) return DSS_E_BADARG;
    if (6 * nb + neq > 64) return DSS_E_UNSUPPORTED;
    if (workspace_bytes < dss_lcp_contact_workspace_bytes(B, nb, neq, maxc, fric_dirs)) return DSS_E_WORKSPACE;
    const size_t lds = lds_bytes(nb, neq, maxc);
    if (lds > 64 * 1024) return DSS_E_UNSUPPORTED;
    if (maxc <= 128) {   // two contacts per lane: IPM state in registers
        const int n = 6 * nb + neq;
#define DSS_LAUNCH_REG(ND_, N_)                                                                                         \
        hipLaunchKernelGGL((lcp_contact_forward_reg_kernel<ND_, N_>), dim3(B), dim3(64), lds, (hipStream_t)stream, Mblk, pvec, \
                           A, bvec, cop, cbody, nc, active, nb, neq, maxc, eps, not_improved_lim, max_iter, x, lam, slack, nu, \
                           iters, status, (double *)workspace)
        if (fric_dirs == 8) { if (n == 54) DSS_LAUNCH_REG(4, 54); else if (n == 18) DSS_LAUNCH_REG(4, 18); else DSS_LAUNCH_REG(4, 0); }
        else { if (n == 54) DSS_LAUNCH_REG(2, 54); else if (n == 18) DSS_LAUNCH_REG(2, 18); else DSS_LAUNCH_REG(2, 0); }
#undef DSS_LAUNCH_REG
    } else if (fric_dirs == 8)
        hipLaunchKernelGGL(lcp_contact_forward_kernel<4>, dim3(B), dim3(64), lds, (hipStream_t)stream, Mblk, pvec, A, bvec,
                           cop, cbody, nc, active, nb, neq, maxc, eps, not_improved_lim, max_iter, x, lam, slack, nu, iters,
                           status, (double *)workspace);
    else
        hipLaunchKernelGGL(lcp_contact_forward_kernel<2>, dim3(B), dim3(64), lds, (hipStream_t)stream, Mblk, pvec, A, bvec,
                           cop, cbody, nc, active, nb, neq, maxc, eps, not_improved_lim, max_iter, x, lam, slack, nu, iters,
                           status, (double *)workspace);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}

int dss_lcp_contact_backward(const double *Mblk, const double *A, const double *cop, const int *cbody, const int *nc,
                             const int *active, int B, int nb, int neq, int maxc, int fric_dirs, const double *x, const double *lam,
                             const double *slack, const double *nu, const double *dl_dx, double *dMblk,
                             double *dpvec, double *dcop, double *dA, double *db, void *stream)
{
    return dss::lcp_contact_backward_rows(Mblk, A, cop, cbody, nc, active, B, nb, neq, maxc, fric_dirs, x, lam, slack, nu, dl_dx, dMblk,
                                          dpvec, dcop, dA, db, 3, nullptr, stream);
}

}  // extern "C"

namespace dss {
int lcp_contact_backward_rows(const double *Mblk, const double *A, const double *cop, const int *cbody, const int *nc,
                              const int *active, int B, int nb, int neq, int maxc, int fric_dirs, const double *x,
                              const double *lam, const double *slack, const double *nu, const double *dl_dx, double *dMblk,
                              double *dpvec, double *dcop, double *dA, double *db, int rows, const int *slot, void *stream)
{
    if (!dims_ok(B, nb, neq, maxc, fric_dirs)) return DSS_E_BADARG;
    if (!Mblk || !cop || !cbody || !nc || !x || !lam || !slack || !dl_dx || !dMblk || !dpvec || !dcop) return DSS_E_BADARG;
    if (neq > 0 && (!A || !nu)) return DSS_E_BADARG;
    if (6 * nb + neq > 64) return DSS_E_UNSUPPORTED;
    const size_t lds = lds_bytes(nb, neq, maxc);
    if (lds > 64 * 1024) return DSS_E_UNSUPPORTED;
    if (fric_dirs == 8)
        hipLaunchKernelGGL(lcp_contact_backward_kernel<4>, dim3(B), dim3(64), lds, (hipStream_t)stream, Mblk, A, cop, cbody,
                           nc, active, nb, neq, maxc, x, lam, slack, nu, dl_dx, dMblk, dpvec, dcop, dA, db, rows, slot, B);
    else
        hipLaunchKernelGGL(lcp_contact_backward_kernel<2>, dim3(B), dim3(64), lds, (hipStream_t)stream, Mblk, A, cop, cbody,
                           nc, active, nb, neq, maxc, x, lam, slack, nu, dl_dx, dMblk, dpvec, dcop, dA, db, rows, slot, B);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}
}  // namespace dss
