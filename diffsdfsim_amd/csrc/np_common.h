// np_common.h -- building blocks shared by the narrow-phase translation units (narrowphase.hip in its lean and full
// compilation, narrowphase_igr.hip): work-item geometry, group-wide scans and reductions, the hull of a normal cluster
// (contacts.py:126-152) and the last two stages of a work item -- thinning (contacts.py:97-158) and output.
// Everything sits in the including file's anonymous namespace; DSS_ALL_SHAPES must be defined before inclusion.
#pragma once
#include <math.h>
#include <stdlib.h>

#include "../../include/diffsdfsim_hip.h"
#include "contact_geom.h"
#include "wave_utils.h"

// 3-D hull of a contact cluster with Qhull's vertex semantics (coincident points are one vertex, points in a triangle of
// others are none).  Needed by every variant: neighbouring faces of an icosphere converge to shared vertices too.
#define DSS_HULL_EXACT 1

namespace {
using namespace dss;

constexpr int NT = 256;
constexpr int MAX_CPT = 8;       // moving candidates per thread of a group: HCAP <= BT * MAX_CPT
constexpr int HULL3_MAX = 48;    // brute-force 3-D hull size limit
// Every item starts on a single wavefront, whatever the size of the mesh it searches: the culling boxes of the 176 k-face
// floor are 688 tests (11 rounds of 64 lanes), after which the item is as small as any other, and a workgroup item keeps
// four wavefronts busy for what is mostly a serial chain.  Items that outgrow the wavefront's scratch are deferred.
constexpr int WAVE_ITEM_MAX_FACES = 1 << 30;

__device__ inline int npairs_of(int nb) { return nb * (nb - 1); }
__device__ inline void pair_of(int dp, int nb, int &a, int &b)
{
    a = dp / (nb - 1);
    const int r = dp % (nb - 1);
    b = r < a ? r : r + 1;
}

struct BodyD {
    BodyG<double> g;
    int mesh, voff, nv, foff, nf;
};

__device__ inline void load_body(const DssWorld &W, int sc, int b, BodyD &o)
{
    const double *ps = W.pose + ((size_t)sc * W.nb + b) * 7;
    for (int i = 0; i < 4; ++i) o.g.q[i] = ps[i];
    for (int i = 0; i < 3; ++i) o.g.pos[i] = ps[4 + i];
    const double *prm = W.shape_prm + ((size_t)sc * W.nb + b) * 3;
#if DSS_ALL_SHAPES
    make_shape(o.g.shape, W.shape_type[(size_t)sc * W.nb + b], prm, W.shape_aux[(size_t)sc * W.nb + b]);
#else
    make_shape(o.g.shape, W.shape_type[(size_t)sc * W.nb + b], prm);
#endif
    // the same for every lane: keep it in scalar registers (frees ~60 VGPRs in the narrow phase)
    for (int i = 0; i < 4; ++i) o.g.q[i] = dss_uniform(o.g.q[i]);
    for (int i = 0; i < 3; ++i) {
        o.g.pos[i] = dss_uniform(o.g.pos[i]);
        o.g.shape.prm[i] = dss_uniform(o.g.shape.prm[i]);
        o.g.shape.hd[i] = dss_uniform(o.g.shape.hd[i]);
    }
    o.g.shape.scale = dss_uniform(o.g.shape.scale);
#if DSS_ALL_SHAPES
    o.g.shape.hr = dss_uniform(o.g.shape.hr);
#endif
    o.g.shape.type = dss_uniform(o.g.shape.type);
#if DSS_ALL_SHAPES
    if (o.g.shape.type == SHAPE_GRID && W.grid_id) {
        const int gi = W.grid_id[(size_t)sc * W.nb + b];
        o.g.shape.grid = W.grid_data + W.grid_off[gi];
        for (int i = 0; i < 3; ++i) o.g.shape.gn[i] = W.grid_dims[3 * gi + i];
    }
#endif
    o.mesh = W.mesh_id[(size_t)sc * W.nb + b];
    o.voff = W.mesh_voff[o.mesh]; o.nv = W.mesh_nv[o.mesh];
    o.foff = W.mesh_foff[o.mesh]; o.nf = W.mesh_nf[o.mesh];
}

// vertex of body `a` (body frame) -> frame of body b, in the reference's order of operations:
// world = R_a v + x_a (get_surface, bodies.py:716-719), then R_b^-1 (world - x_b) (contacts.py:42)
__device__ inline void to_frame(const BodyG<double> &a, const BodyG<double> &b, const double *v, double *o)
{
    double w[3], r[3];
    quat_apply(a.q, v, w);
    for (int i = 0; i < 3; ++i) r[i] = (w[i] + a.pos[i]) - b.pos[i];
    quat_apply_inv(b.q, r, o);
}


// axis-aligned box, in the body frame of `a`, that contains body b's query cube [-s,s]^3 (+ margin)
struct Region { double c[3], e[3]; };
__device__ inline void region_of(const BodyG<double> &a, const BodyG<double> &b, double margin, Region &r)
{
    double Ra[9], Rb[9];
    quat_to_mat(a.q, Ra);
    quat_to_mat(b.q, Rb);
    const double d[3] = {b.pos[0] - a.pos[0], b.pos[1] - a.pos[1], b.pos[2] - a.pos[2]};
    const double s = b.shape.scale;
    for (int i = 0; i < 3; ++i) {
        r.c[i] = Ra[i] * d[0] + Ra[3 + i] * d[1] + Ra[6 + i] * d[2];          // R_a^T (x_b - x_a)
        double e = 0.0;
        for (int j = 0; j < 3; ++j) e += fabs(Ra[i] * Rb[j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j]);   // |R_a^T R_b|
        r.e[i] = s * e * (1.0 + 1e-9) + margin;
    }
}
__device__ inline bool box_hits(const Region &r, const double *bx)
{
    for (int i = 0; i < 3; ++i)
        if (bx[i] > r.c[i] + r.e[i] || bx[3 + i] < r.c[i] - r.e[i]) return false;
    return true;
}

// ---- workgroup scratch ------------------------------------------------------------------------
// A work item (scene, directed pair a->b) is processed by a GROUP of threads: a whole 256-thread workgroup when
// a's mesh is big (the 176 k-face floor), a single wavefront otherwise.  Most of the item is a serial chain
// (Frank-Wolfe iterations of a handful of movers, greedy clustering, gift wrapping), so four independent
// wavefronts per workgroup keep four times as many chains in flight; the wave flavour needs no s_barrier at all.
//   BT    threads of the group             HCAP  capacity of the mover list / of one normal cluster
//   CHCAP runs of 256 faces the barrier-free scan can hold (176 k-face floor = 688 runs)
template <int BT_, int HCAP_, int CHCAP_> struct Group {
    static constexpr int BT = BT_, HCAP = HCAP_, CHCAP = CHCAP_, NW = BT_ / 64;
    __device__ static inline int tid() { return BT_ == 64 ? (int)(threadIdx.x & 63) : (int)threadIdx.x; }
    __device__ static inline void sync() { if (BT_ == 64) dss_wave_sync(); else __syncthreads(); }
    __device__ static inline int any(int x) { if (BT_ == 64) return __ballot(x) != 0ull; else return __syncthreads_or(x); }
};
using BlockGroup = Group<256, 1024, 704>;
#if defined(DSS_NP_EXP_HCAP)
using WaveGroup = Group<64, DSS_NP_EXP_HCAP, 32>;
#else
using WaveGroup = Group<64, 384, 32>;
#endif

template <class G> struct ScratchT {
    int wave_tot[G::NW];
    int woff[G::BT == 64 ? 1 : G::CHCAP * 4];   // (workgroup scan only)
    int vote[2][G::NW];
    int red_i[G::BT];
    double red_d[G::BT];
    double hp[3 * G::HCAP];   // cluster points for the hull
    int hidx[G::HCAP];
    unsigned char hflag[G::HCAP];
    unsigned char cst[G::HCAP];   // filter state of a contact: 0 unassigned, 1 clustered, 2 kept, 255 no normal
};

// ordered compaction: returns this thread's output slot (or -1) and updates the running count
template <class G> __device__ inline int compact_slot(int flag, int &count, ScratchT<G> &S)
{
    const int tid = G::tid(), lane = tid & 63, wv = tid >> 6;
    const unsigned long long m = __ballot(flag);
    const int pre = __popcll(m & ((1ull << lane) - 1ull)), tot = __popcll(m);
    if (G::BT == 64) { const int off1 = count; count += tot; return flag ? off1 + pre : -1; }
    if (lane == 0) S.wave_tot[wv] = tot;
    G::sync();
    int off = count, all = 0;
    for (int w = 0; w < G::BT / 64; ++w) { if (w < wv) off += S.wave_tot[w]; all += S.wave_tot[w]; }
    G::sync();
    count += all;
    return flag ? off + pre : -1;
}

// block arg-min over (key, index) with lowest index on ties; returns the index (or -1 if none valid)
// value of lane 0, for every lane (the xor butterfly leaves the same set summed in every lane, but only lane 0's
// order of additions is that of the LDS tree the workgroup flavour uses)
__device__ inline double wave_first(double x)
{
#if defined(DSS_EMU)
    return __shfl(x, 0, 64);
#else
    return dss_uniform(x);
#endif
}
template <class G> __device__ inline int block_argmin(double key, int idx, ScratchT<G> &S)
{
    if (G::BT == 64) {   // one wavefront: shuffles, no LDS round trips; (key, index) minimum is order independent
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ok = __shfl_xor(key, o, 64);
            const int oi = __shfl_xor(idx, o, 64);
            if (oi >= 0 && (idx < 0 || ok < key || (ok == key && oi < idx))) { key = ok; idx = oi; }
        }
        return idx;
    }
    const int tid = G::tid();
    S.red_d[tid] = key; S.red_i[tid] = idx;
    G::sync();
    for (int s = G::BT / 2; s > 0; s >>= 1) {
        if (tid < s) {
            const double ok = S.red_d[tid + s]; const int oi = S.red_i[tid + s];
            const int mi = S.red_i[tid];
            if (oi >= 0 && (mi < 0 || ok < S.red_d[tid] || (ok == S.red_d[tid] && oi < mi))) { S.red_d[tid] = ok; S.red_i[tid] = oi; }
        }
        G::sync();
    }
    const int r = S.red_i[0];
    G::sync();
    return r;
}
template <class G> __device__ inline double block_max(double v, ScratchT<G> &S)
{
    if (G::BT == 64) return wave_max_dpp(v);
    const int tid = G::tid();
    S.red_d[tid] = v;
    G::sync();
    for (int s = G::BT / 2; s > 0; s >>= 1) { if (tid < s) S.red_d[tid] = fmax(S.red_d[tid], S.red_d[tid + s]); G::sync(); }
    const double r = S.red_d[0];
    G::sync();
    return r;
}
template <class G> __device__ inline double block_sum(double v, ScratchT<G> &S)
{
    if (G::BT == 64) return wave_first(wave_sum(v));   // same association as the tree below, bit for bit
    const int tid = G::tid();
    S.red_d[tid] = v;
    G::sync();
    for (int s = G::BT / 2; s > 0; s >>= 1) { if (tid < s) S.red_d[tid] += S.red_d[tid + s]; G::sync(); }
    const double r = S.red_d[0];
    G::sync();
    return r;
}

// ---- hull of one normal cluster (points in S.hp, m of them); marks S.hflag ---------------------
// Mirrors the fall-back ladder of contacts.py:126-152: 3-D hull; if Qhull would reject the input as
// flat drop the coordinate of least variance and retry in 2-D; then 1-D min/max.
// where a cluster's points and keep-flags live while its hull is taken: in LDS (clusters of up to HCAP points) or, for the
// clusters a level-set mesh resting flat on a neighbour produces (every face of the resting side), in the group's global
// candidate scratch (rows 3-6 of cand_buf, dead after the contact geometry stage)
template <class G> struct HullLds {
    ScratchT<G> *S;
    __device__ inline double hp(int k, int d) const { return S->hp[3 * k + d]; }
    __device__ inline int getf(int k) const { return S->hflag[k]; }
    __device__ inline void setf(int k, int v) const { S->hflag[k] = (unsigned char)v; }
};
struct HullGlobal {
    double *cb; int mc;
    __device__ inline double hp(int k, int d) const { return cb[(size_t)(3 + d) * mc + k]; }
    __device__ inline int getf(int k) const { return (int)cb[(size_t)6 * mc + k]; }
    __device__ inline void setf(int k, int v) const { cb[(size_t)6 * mc + k] = (double)v; }
};


#if DSS_ALL_SHAPES
// ---- 3-D hull by gift wrapping (workgroup flavour, clusters beyond the brute-force limit) ----------------------------
// A level-set body at rest on a side with a rounded rim gives one normal cluster of thousands of contact points -- the
// resting face plus the first rows of the rim a fraction of a millimetre above it -- of which Qhull (contacts.py:126-152)
// keeps the few dozen that are vertices of the 3-D hull.  Wrapping finds exactly those: around every directed hull edge
// a -> b the next face is (a, b, c) with every point on or behind its plane.  Ties among points IN that plane (a facet
// with hundreds of coplanar points) are broken like the 2-D wrap does it: the next vertex of the facet's polygon after b,
// the farthest of collinear ones -- so the facet is fanned out over its polygon's vertices only and no interior or
// mid-edge point ever becomes a vertex.  Coincident points are one vertex (the lowest index wins every tie).
// Work per face: one pass over the points by the whole workgroup + an LDS tree; bookkeeping (the list of directed edges)
// in the scan's LDS words, which are free by now.  Returns false (nothing flagged) if the wrap cannot be trusted: no
// unique extreme start point in any of the probe directions, or more faces than the edge list holds.
template <class P_> struct WrapEdge {
    double a[3], e[3], le2;
    double nx[3];   // e x (x - a), x = the third vertex of the face this edge was found in (hasx), which lies behind the new face
    int ia, ib, ix; // ia < 0: `a` is a virtual point (the start); ix < 0: no neighbouring face yet
};
template <class P_> __device__ inline bool wrap_valid(const P_ &P, const WrapEdge<P_> &E, int c, double tolf)
{
    if (c == E.ia || c == E.ib || c == E.ix) return false;
    const double u[3] = {P.hp(c, 0) - E.a[0], P.hp(c, 1) - E.a[1], P.hp(c, 2) - E.a[2]};
    double m[3];
    cross(E.e, u, m);
    return m[0] * m[0] + m[1] * m[1] + m[2] * m[2] > tolf * tolf * E.le2;      // off the edge's line (and off a and b)
}
// does candidate c2 take the place of c1 (both valid)
template <class P_> __device__ inline bool wrap_better(const P_ &P, const WrapEdge<P_> &E, int c1, int c2, double tolf, double dtol2)
{
    if (c2 < 0) return false;
    if (c1 < 0) return true;
    const double u1[3] = {P.hp(c1, 0) - E.a[0], P.hp(c1, 1) - E.a[1], P.hp(c1, 2) - E.a[2]};
    const double u2[3] = {P.hp(c2, 0) - E.a[0], P.hp(c2, 1) - E.a[1], P.hp(c2, 2) - E.a[2]};
    double m1[3];
    cross(E.e, u1, m1);
    const double lm = sqrt(m1[0] * m1[0] + m1[1] * m1[1] + m1[2] * m1[2]);
    const double s = (m1[0] * u2[0] + m1[1] * u2[1] + m1[2] * u2[2]) / lm;      // c2 above (+) / below the plane (a, b, c1)
    if (s > tolf) return true;
    if (s < -tolf) return false;
    // In the plane.  An edge that is a DIAGONAL of a coplanar facet (the fan over a facet's polygon produces them) has
    // coplanar points on both sides of its line: those on the side of the face it was found in are behind it.
    if (E.ix >= 0) {
        double m2[3];
        cross(E.e, u2, m2);
        if (m1[0] * m2[0] + m1[1] * m2[1] + m1[2] * m2[2] < 0.0) {       // c1 and c2 on opposite sides of the edge's line
            const double b1 = m1[0] * E.nx[0] + m1[1] * E.nx[1] + m1[2] * E.nx[2];
            return b1 > 0.0;      // c1 lies on x's side: c2 takes its place
        }
    }
    // seen from outside the facet's polygon runs counter-clockwise, a -> b -> c with every point left of b -> c
    const double v1[3] = {u1[0] - E.e[0], u1[1] - E.e[1], u1[2] - E.e[2]}, v2[3] = {u2[0] - E.e[0], u2[1] - E.e[1], u2[2] - E.e[2]};
    double x[3];
    cross(v1, v2, x);
    const double cr = (x[0] * m1[0] + x[1] * m1[1] + x[2] * m1[2]) / lm;
    const double lb = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2], lq = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
    const double c2s = cr * cr, t2 = dtol2 * fmax(lb, lq);     // the 2-D wrap's (Qhull's) measure of "collinear"
    if (cr < 0.0 && c2s > t2) return true;
    if (c2s <= t2) return lq > lb || (lq == lb && c2 < c1);
    return false;
}
template <class G, class P_> __device__ int wrap_next(ScratchT<G> &S, const P_ &P, const WrapEdge<P_> &E, int m, double tolf, double dtol2)
{
    const int tid = G::tid();
    int best = -1;
    for (int k = tid; k < m; k += G::BT)
        if (wrap_valid(P, E, k, tolf) && wrap_better(P, E, best, k, tolf, dtol2)) best = k;
    S.red_i[tid] = best;
    G::sync();
    for (int s = G::BT / 2; s > 0; s >>= 1) {
        if (tid < s) { const int a = S.red_i[tid], b = S.red_i[tid + s]; if (wrap_better(P, E, a, b, tolf, dtol2)) S.red_i[tid] = b; }
        G::sync();
    }
    const int r = S.red_i[0];
    G::sync();
    return r;
}
template <class G, class P_> __device__ bool hull3_wrap(ScratchT<G> &S, const P_ &P, int m, double amax, double tolf, double dtol2)
{
    constexpr int EMAX = (G::CHCAP * 4) / 4;       // directed edges the list holds (four int arrays in S.woff)
    if (EMAX < 96) return false;
    const int tid = G::tid();
    int *eu = S.woff, *ev = S.woff + EMAX, *est = S.woff + 2 * EMAX, *ex = S.woff + 3 * EMAX;   // est: 0 = needs its face, 1 = has it; ex: third vertex of the face found
    // a start vertex: the extreme point of a direction in which it is the only one (no facet or edge perpendicular to it)
    const double probes[3][3] = {{0.5411961001, 0.6363961031, 0.4209517757}, {-0.3826834324, 0.5879378012, 0.7126966451},
                                 {0.7071067812, -0.4539904997, 0.5420261274}};
    int i0 = -1;
    double gdir[3] = {0, 0, 0};
    for (int t = 0; t < 3 && i0 < 0; ++t) {
        const double *g = probes[t];
        double key = INFINITY; int ki = -1;
        for (int k = tid; k < m; k += G::BT) {
            const double v = g[0] * P.hp(k, 0) + g[1] * P.hp(k, 1) + g[2] * P.hp(k, 2);
            if (v < key) { key = v; ki = k; }
        }
        const int c = block_argmin(key, ki, S);
        const double vmin = g[0] * P.hp(c, 0) + g[1] * P.hp(c, 1) + g[2] * P.hp(c, 2);
        int other = 0;
        for (int k = tid; k < m; k += G::BT) {
            const double v = g[0] * P.hp(k, 0) + g[1] * P.hp(k, 1) + g[2] * P.hp(k, 2);
            const double d0 = P.hp(k, 0) - P.hp(c, 0), d1 = P.hp(k, 1) - P.hp(c, 1), d2 = P.hp(k, 2) - P.hp(c, 2);
            other |= (v <= vmin + tolf) && (d0 * d0 + d1 * d1 + d2 * d2 > tolf * tolf);
        }
        if (!G::any(other)) { i0 = c; for (int d = 0; d < 3; ++d) gdir[d] = g[d]; }
    }
    if (i0 < 0) return false;
    // first edge: wrap around a virtual line through the start vertex inside its supporting plane
    WrapEdge<P_> E;
    {
        const double ex[3] = {1.0, 0.0, 0.0};
        double d[3];
        cross(gdir, ex, d);
        const double ld = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), L = 1.0 + amax;
        for (int k = 0; k < 3; ++k) { E.e[k] = d[k] / ld * L; E.a[k] = P.hp(i0, k) - E.e[k]; }
        E.le2 = L * L; E.ia = -1; E.ib = i0; E.ix = -1;
        E.nx[0] = E.nx[1] = E.nx[2] = 0.0;
    }
    const int i1 = wrap_next(S, P, E, m, tolf, dtol2);
    if (i1 < 0) return false;
    if (tid == 0) { eu[0] = i0; ev[0] = i1; est[0] = 0; ex[0] = -1; eu[1] = i1; ev[1] = i0; est[1] = 0; ex[1] = -1; P.setf(i0, 1); P.setf(i1, 1); S.wave_tot[0] = 2; }
    G::sync();
    int ne = 2;
    // directed edge a -> b of the current facet gets its face; its reverse needs one unless it is listed already
    auto settle = [&](int a, int b, int third) {
        if (tid < 2) S.red_i[tid] = -1;
        G::sync();
        for (int e = tid; e < ne; e += G::BT) {
            if (eu[e] == a && ev[e] == b) S.red_i[0] = e;
            if (eu[e] == b && ev[e] == a) S.red_i[1] = e;
        }
        G::sync();
        if (tid == 0) {
            int n2 = ne;
            if (S.red_i[0] >= 0) est[S.red_i[0]] = 1;
            else { if (n2 < EMAX) { eu[n2] = a; ev[n2] = b; est[n2] = 1; ex[n2] = -1; } ++n2; }
            if (S.red_i[1] < 0) { if (n2 < EMAX) { eu[n2] = b; ev[n2] = a; est[n2] = 0; ex[n2] = third; } ++n2; }
            S.wave_tot[0] = n2;
        }
        G::sync();
        ne = S.wave_tot[0];
        G::sync();
    };
    for (int q = 0; q < ne; ++q) {
        if (est[q]) continue;           // (uniform: LDS word read by every thread)
        const int u = eu[q], v = ev[q];
        for (int k = 0; k < 3; ++k) { E.a[k] = P.hp(u, k); E.e[k] = P.hp(v, k) - E.a[k]; }
        E.le2 = E.e[0] * E.e[0] + E.e[1] * E.e[1] + E.e[2] * E.e[2]; E.ia = u; E.ib = v; E.ix = ex[q];
        if (E.ix >= 0) {
            const double xr[3] = {P.hp(E.ix, 0) - E.a[0], P.hp(E.ix, 1) - E.a[1], P.hp(E.ix, 2) - E.a[2]};
            cross(E.e, xr, E.nx);
        }
        const int w = wrap_next(S, P, E, m, tolf, dtol2);
        if (w < 0) return false;
        // The facet through (u, v, w) as a WHOLE polygon: from v -> w on, the next vertex is the next of the 2-D wrap among the
        // points in the facet's plane, until the walk is back at u.  Every boundary edge gets its face at once and no
        // diagonal is ever created: a fan over the polygon would depend on the edge the facet is entered through, and two
        // entries (its neighbours are found in any order) would triangulate it in two incompatible ways.
        double nrm[3];
        {
            const double uw[3] = {P.hp(w, 0) - E.a[0], P.hp(w, 1) - E.a[1], P.hp(w, 2) - E.a[2]};
            cross(E.e, uw, nrm);
            const double ln = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
            for (int k = 0; k < 3; ++k) nrm[k] /= ln;
        }
        const double org[3] = {E.a[0], E.a[1], E.a[2]};
        settle(u, v, w);
        int pa = v, pb = w;       // the edge being added; then the walk goes on from pb
        for (int step = 0; step < m; ++step) {
            if (tid == 0) P.setf(pb, 1);
            settle(pa, pb, u == pa ? w : u);      // (third vertex for the reverse edge: any vertex of this facet off the edge)
            if (ne > EMAX) return false;
            if (pb == u) break;
            // next vertex after pb: among the points in the plane, the one with every other to the left of pb -> c
            const double bx = P.hp(pb, 0), by = P.hp(pb, 1), bz = P.hp(pb, 2);
            auto better = [&](int c1, int c2) -> bool {
                if (c2 < 0) return false;
                if (c1 < 0) return true;
                const double v1[3] = {P.hp(c1, 0) - bx, P.hp(c1, 1) - by, P.hp(c1, 2) - bz}, v2[3] = {P.hp(c2, 0) - bx, P.hp(c2, 1) - by, P.hp(c2, 2) - bz};
                double x[3];
                cross(v1, v2, x);
                const double cr = x[0] * nrm[0] + x[1] * nrm[1] + x[2] * nrm[2];
                const double lb = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2], lq = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
                const double c2s = cr * cr, t2 = dtol2 * fmax(lb, lq);
                if (cr < 0.0 && c2s > t2) return true;
                if (c2s <= t2) return lq > lb || (lq == lb && c2 < c1);
                return false;
            };
            int best = -1;
            for (int k = tid; k < m; k += G::BT) {
                if (k == pb) continue;
                const double d[3] = {P.hp(k, 0) - org[0], P.hp(k, 1) - org[1], P.hp(k, 2) - org[2]};
                if (fabs(d[0] * nrm[0] + d[1] * nrm[1] + d[2] * nrm[2]) > tolf) continue;          // not in the facet's plane
                const double r[3] = {P.hp(k, 0) - bx, P.hp(k, 1) - by, P.hp(k, 2) - bz};
                if (!(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] > tolf * tolf)) continue;             // pb itself or a duplicate of it
                if (better(best, k)) best = k;
            }
            S.red_i[tid] = best;
            G::sync();
            for (int s2 = G::BT / 2; s2 > 0; s2 >>= 1) {
                if (tid < s2) { const int a = S.red_i[tid], b = S.red_i[tid + s2]; if (better(a, b)) S.red_i[tid] = b; }
                G::sync();
            }
            int nx2 = S.red_i[0];
            G::sync();
            if (nx2 < 0) return false;
            {   // coincident with the start (a duplicate of u with another index): the polygon is closed
                const double d[3] = {P.hp(nx2, 0) - org[0], P.hp(nx2, 1) - org[1], P.hp(nx2, 2) - org[2]};
                if (!(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] > tolf * tolf)) nx2 = u;
            }
            pa = pb; pb = nx2;
            if (step + 1 == m) return false;      // (no closed polygon: not to be trusted)
        }
    }
    return true;
}
#endif

// Returns 2 (lean variant) if the cluster has more than 48 distinct non-coplanar points -- every one of them is kept and the
// caller flags the scene; 1 if a wavefront-sized group has to hand the item to a workgroup (full variant: a cluster of more than 48 distinct
// non-coplanar points gets the gift-wrapped hull, which keeps its edge list in the workgroup's LDS), else 0.
template <class G, class P_> __device__ int cluster_hull(ScratchT<G> &S, P_ P, int m, double eps)
{
    const int tid = G::tid();
    for (int k = tid; k < m; k += G::BT) P.setf(k, 0);
    G::sync();
    if (m == 1) { if (tid == 0) P.setf(0, 1); G::sync(); return 0; }
    // per-coordinate mean / unbiased variance (torch.var)
    double mean[3], var[3], amax = 0.0;
    for (int d = 0; d < 3; ++d) {
        double acc = 0.0, mx = 0.0;
        for (int k = tid; k < m; k += G::BT) { acc += P.hp(k, d); mx = fmax(mx, fabs(P.hp(k, d))); }
        mean[d] = block_sum(acc, S) / m;
        amax = fmax(amax, block_max(mx, S));
        acc = 0.0;
        for (int k = tid; k < m; k += G::BT) { const double t = P.hp(k, d) - mean[d]; acc += t * t; }
        var[d] = block_sum(acc, S) / (m - 1);
    }
    const double tolf = 1e-12 * (1.0 + amax);
    // distance tolerance of the 2-D hull: Qhull merges a vertex that clears its neighbours' edge by less than ~6e-15 of the
    // extent (its `_one-merge`); a box that has turned by 4e-15 rad puts its mid-edge contact points 2e-15 off the edge
    // (dropped there), the smallest excursion seen kept is 4e-11
    const double dtol = 2e-14 * (1.0 + amax), dtol2 = dtol * dtol;
    // farthest point B from A = point 0, then C farthest from line AB
    const double A[3] = {P.hp(0, 0), P.hp(0, 1), P.hp(0, 2)};
    double key = -1.0; int ki = -1;
    for (int k = tid; k < m; k += G::BT) {
        const double d0 = P.hp(k, 0) - A[0], d1 = P.hp(k, 1) - A[1], d2 = P.hp(k, 2) - A[2];
        const double dd = d0 * d0 + d1 * d1 + d2 * d2;
        if (dd > key) { key = dd; ki = k; }
    }
    const int iB = block_argmin(-key, ki, S);
    double ab[3] = {P.hp(iB, 0) - A[0], P.hp(iB, 1) - A[1], P.hp(iB, 2) - A[2]};
    const double lab = sqrt(ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2]);
    bool flat3 = (m < 4) || !(lab > tolf), line = !(lab > tolf);
    double nrm[3] = {0, 0, 0};
    if (!line) {
        key = -1.0; ki = -1;
        for (int k = tid; k < m; k += G::BT) {
            const double d[3] = {P.hp(k, 0) - A[0], P.hp(k, 1) - A[1], P.hp(k, 2) - A[2]};
            double c[3];
            cross(ab, d, c);
            const double dd = (c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
            if (dd > key) { key = dd; ki = k; }
        }
        const int iC = block_argmin(-key, ki, S);
        const double ac[3] = {P.hp(iC, 0) - A[0], P.hp(iC, 1) - A[1], P.hp(iC, 2) - A[2]};
        cross(ab, ac, nrm);
        const double ln = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
        if (!(ln / lab > tolf)) { line = true; flat3 = true; }
        else {
            for (int d = 0; d < 3; ++d) nrm[d] /= ln;
            double mx = -INFINITY, mn = INFINITY; int imx = -1, imn = -1;
            for (int k = tid; k < m; k += G::BT) {
                const double sd = nrm[0] * (P.hp(k, 0) - A[0]) + nrm[1] * (P.hp(k, 1) - A[1]) + nrm[2] * (P.hp(k, 2) - A[2]);
                if (sd > mx) { mx = sd; imx = k; }
                if (sd < mn) { mn = sd; imn = k; }
            }
            const int gmx = block_argmin(-mx, imx, S), gmn = block_argmin(mn, imn, S);
            const double dmx = nrm[0] * (P.hp(gmx, 0) - A[0]) + nrm[1] * (P.hp(gmx, 1) - A[1]) + nrm[2] * (P.hp(gmx, 2) - A[2]);
            const double dmn = nrm[0] * (P.hp(gmn, 0) - A[0]) + nrm[1] * (P.hp(gmn, 1) - A[1]) + nrm[2] * (P.hp(gmn, 2) - A[2]);
            const double thick = fmax(fabs(dmx), fabs(dmn));
            if (!(thick > tolf)) flat3 = true;
            else if (thick <= 1e-6 * (1.0 + amax) && m >= 4) {
                // a sliver: its 3-D hull is the 2-D hull of the projection plus the points that stick out of
                // the plane (Qhull keeps those as vertices; anything flatter than round-off it rejects outright)
                flat3 = true;
                if (tid == 0) { if (fabs(dmx) > tolf) P.setf(gmx, 2); if (fabs(dmn) > tolf) P.setf(gmn, 2); }   // 2 = kept, not a visited hull vertex
                G::sync();
            }
        }
    }
#if DSS_HULL_EXACT
#if DSS_ALL_SHAPES
    // workgroup flavour, more points than the pairwise duplicate search below is meant for: gift wrapping
    if (!flat3 && G::BT != 64 && m > 2048) {
        if (hull3_wrap(S, P, m, amax, tolf, dtol2)) return 0;
        for (int k = tid; k < m; k += G::BT) P.setf(k, 0);      // could not be trusted: every point is kept (below)
        G::sync();
    }
#endif
    if (!flat3 && m > 2048) {   // beyond what the pairwise duplicate search below is meant for: keep every point
        for (int k = tid; k < m; k += G::BT) P.setf(k, 1);
        G::sync();
        return 0;
    }
    if (!flat3) {
        // Coincident points (candidates of neighbouring faces that converged to a shared mesh vertex -- the rule on a
        // level-set mesh) are one hull vertex to Qhull: the first of each group stands for it.  3 = duplicate.
        for (int k = tid; k < m; k += G::BT) {
            int dup = 0;
            for (int j = 0; j < k && !dup; ++j) {
                const double d0 = P.hp(k, 0) - P.hp(j, 0), d1 = P.hp(k, 1) - P.hp(j, 1), d2 = P.hp(k, 2) - P.hp(j, 2);
                dup = !(d0 * d0 + d1 * d1 + d2 * d2 > tolf * tolf);
            }
            if (dup) P.setf(k, 3);
        }
        G::sync();
        static_assert(HULL3_MAX <= 64, "the list of distinct points lives in red_i");
        if (tid == 0) {
            int mu = 0;
            for (int k = 0; k < m; ++k) if (P.getf(k) != 3) { if (mu < HULL3_MAX) S.red_i[mu] = k; ++mu; }
            S.wave_tot[0] = mu;
        }
        G::sync();
        const int mu = S.wave_tot[0];
        G::sync();
        if (mu > HULL3_MAX) {
            // Beyond the brute-force limit: a superset of the hull's vertices -- every distinct point, except (full
            // variant, up to 512 points) those that lie on the segment between two other points.  That is what a curved
            // level-set surface in line contact needs (a cylinder lying on the floor: rows of collinear contact points
            // along its length, of which only the two ends of a row are vertices).
#if DSS_ALL_SHAPES
            // the distinct points, listed once (workgroup: in the scan's LDS words, free by now; a wavefront's cluster is
            // small enough to walk with its duplicates)
            // the workgroup flavour takes the real hull (gift wrapping); the segment filter below remains for the wavefront
            // flavour (clusters of up to 384 points) and as the fall-back should the wrap not close
            if (G::BT == 64) return 1;      // (uniform: every lane sees the same count)
            if (G::BT != 64) {
                if (hull3_wrap(S, P, m, amax, tolf, dtol2)) {
                    for (int k = tid; k < m; k += G::BT) if (P.getf(k) == 3) P.setf(k, 0);
                    G::sync();
                    return 0;
                }
                for (int k = tid; k < m; k += G::BT) if (P.getf(k) == 1) P.setf(k, 0);
                G::sync();
            }
            constexpr int UCAP = G::BT == 64 ? 1 : 512;
            const bool listed = G::BT != 64 && mu <= UCAP;
            if (listed && tid == 0) { int u = 0; for (int k = 0; k < m; ++k) if (P.getf(k) != 3) S.woff[u++] = k; }
            G::sync();
            const bool thin = listed || (G::BT == 64 && mu <= 512);
            const int nu = listed ? mu : m;
#else
            const bool thin = false, listed = false;
            const int nu = m;
#endif
            for (int q = tid; q < m; q += G::BT) {
                int keep = P.getf(q) != 3;
                if (keep && thin) {
                    const double qx = P.hp(q, 0), qy = P.hp(q, 1), qz = P.hp(q, 2);
                    for (int ia = 0; ia < nu && keep; ++ia) {
                        const int a = listed ? S.woff[ia] : ia;
                        if (a == q || P.getf(a) == 3) continue;
                        const double ax = P.hp(a, 0) - qx, ay = P.hp(a, 1) - qy, az = P.hp(a, 2) - qz;
                        for (int ib = ia + 1; ib < nu; ++ib) {
                            const int b = listed ? S.woff[ib] : ib;
                            if (b == q || P.getf(b) == 3) continue;
                            const double bx = P.hp(b, 0) - qx, by = P.hp(b, 1) - qy, bz = P.hp(b, 2) - qz;
                            if (!(ax * bx + ay * by + az * bz < 0.0)) continue;          // q is not between a and b
                            const double cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
                            const double ex = bx - ax, ey = by - ay, ez = bz - az;
                            // distance of q from the line a-b, squared: |a x b|^2 / |b - a|^2
                            if (cx * cx + cy * cy + cz * cz <= tolf * tolf * (ex * ex + ey * ey + ez * ez)) { keep = 0; break; }
                        }
                    }
                }
                if (P.getf(q) != 3) P.setf(q, keep ? 1 : 4);      // other threads only ask whether a flag is 3
            }
            G::sync();
            for (int k = tid; k < m; k += G::BT) { const int f = P.getf(k); P.setf(k, f == 1 ? 1 : 0); }
            G::sync();
            // (the lean variant has no exact hull beyond the brute-force limit: what it keeps is a superset, and the caller
            // reports that the scene needs the full kernel variants)
            return DSS_ALL_SHAPES ? 0 : 2;
        }
        // supporting-plane test over all triples of distinct points.  A point that lies in the triangle of three others
        // (inside it or on one of its edges) is no hull vertex, whatever planes it helps to support: Qhull reports the
        // extreme points only, a level-set mesh puts rows of equidistant vertices on every straight edge.
        for (int q = tid; q < mu; q += G::BT) S.red_d[q] = 0.0;      // 1 = lies in a triangle of other points
        G::sync();
        const int ntri = mu * mu * mu;
        for (int e = tid; e < ntri; e += G::BT) {
            const int iu = e / (mu * mu), ju = (e / mu) % mu, ku = e % mu;
            if (!(iu < ju && ju < ku)) continue;
            const int i = S.red_i[iu], j = S.red_i[ju], k = S.red_i[ku];
#else
    if (!flat3) {
        if (m > HULL3_MAX) {  // beyond the brute-force limit: keep every point (superset of the hull)
            for (int k = tid; k < m; k += G::BT) P.setf(k, 1);
            G::sync();
            return 0;
        }
        // supporting-plane test over all triples
        const int ntri = m * m * m;
        for (int e = tid; e < ntri; e += G::BT) {
            const int i = e / (m * m), j = (e / m) % m, k = e % m;
            if (!(i < j && j < k)) continue;
#endif
            const double u[3] = {P.hp(j, 0) - P.hp(i, 0), P.hp(j, 1) - P.hp(i, 1), P.hp(j, 2) - P.hp(i, 2)};
            const double v[3] = {P.hp(k, 0) - P.hp(i, 0), P.hp(k, 1) - P.hp(i, 1), P.hp(k, 2) - P.hp(i, 2)};
            double n[3];
            cross(u, v, n);
            const double ln = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            if (!(ln > 1e-14 * (1.0 + amax) * (1.0 + amax))) continue;
            int pos = 0, neg = 0;
            for (int q = 0; q < m; ++q) {
                const double sd = (n[0] * (P.hp(q, 0) - P.hp(i, 0)) + n[1] * (P.hp(q, 1) - P.hp(i, 1)) + n[2] * (P.hp(q, 2) - P.hp(i, 2))) / ln;
                pos |= sd > tolf; neg |= sd < -tolf;
            }
#if DSS_HULL_EXACT
            for (int qu = 0; qu < mu; ++qu) {
                const int q = S.red_i[qu];
                if (q == i || q == j || q == k) continue;
                const double w[3] = {P.hp(q, 0) - P.hp(i, 0), P.hp(q, 1) - P.hp(i, 1), P.hp(q, 2) - P.hp(i, 2)};
                if (fabs(n[0] * w[0] + n[1] * w[1] + n[2] * w[2]) / ln > tolf) continue;      // not in the triangle's plane
                // barycentric coordinates of q in (i, j, k): areas of the sub-triangles over the area of the triangle
                double c1[3], c2[3];
                cross(w, v, c1);
                cross(u, w, c2);
                const double bj = (c1[0] * n[0] + c1[1] * n[1] + c1[2] * n[2]) / (ln * ln);
                const double bk = (c2[0] * n[0] + c2[1] * n[1] + c2[2] * n[2]) / (ln * ln);
                if (bj >= -1e-9 && bk >= -1e-9 && 1.0 - bj - bk >= -1e-9) S.red_d[qu] = 1.0;
            }
#endif
            if (!(pos && neg)) { P.setf(i, 1); P.setf(j, 1); P.setf(k, 1); }
        }
        G::sync();
#if DSS_HULL_EXACT
        for (int k = tid; k < m; k += G::BT) if (P.getf(k) == 3) P.setf(k, 0);
        for (int qu = tid; qu < mu; qu += G::BT) if (S.red_d[qu] != 0.0) P.setf(S.red_i[qu], 0);
        G::sync();
#endif
        return 0;
    }
    // ---- 2-D: drop the coordinate of least variance (first index on ties, torch.argmin) ---------
    int drop = 0;
    for (int d = 1; d < 3; ++d) if (var[d] < var[drop]) drop = d;
    if (var[0] != var[0]) drop = 0;
    const int c0 = drop == 0 ? 1 : 0, c1 = drop == 2 ? 1 : 2;
    bool collinear = (m < 3);
    int iS = -1;
    if (!collinear) {
        // start: lexicographic minimum.  "Equal" first coordinates are equal to round-off: contact points along a box edge
        // are collinear in the body frame and pick up 1e-17 of noise in the rotation to the world frame; taking the
        // bare minimum would start (and keep) a point from the middle of that edge
        double k0 = INFINITY; int k0i = -1;
        for (int k = tid; k < m; k += G::BT) if (P.hp(k, c0) < k0) { k0 = P.hp(k, c0); k0i = k; }
        const int i0 = block_argmin(k0, k0i, S);
        const double x0 = P.hp(i0, c0);
        k0 = INFINITY; k0i = -1;
        const double xtol = dtol;
        for (int k = tid; k < m; k += G::BT) if (P.hp(k, c0) <= x0 + xtol && P.hp(k, c1) < k0) { k0 = P.hp(k, c1); k0i = k; }
        iS = block_argmin(k0, k0i, S);
        // collinearity: farthest point from the start, then max distance to that line
        key = -1.0; ki = -1;
        for (int k = tid; k < m; k += G::BT) {
            const double d0 = P.hp(k, c0) - P.hp(iS, c0), d1 = P.hp(k, c1) - P.hp(iS, c1);
            if (d0 * d0 + d1 * d1 > key) { key = d0 * d0 + d1 * d1; ki = k; }
        }
        const int iF = block_argmin(-key, ki, S);
        const double e0 = P.hp(iF, c0) - P.hp(iS, c0), e1 = P.hp(iF, c1) - P.hp(iS, c1);
        const double le = sqrt(e0 * e0 + e1 * e1);
        if (!(le > tolf)) collinear = true;
        else {
            double mx = 0.0;
            for (int k = tid; k < m; k += G::BT)
                mx = fmax(mx, fabs(e0 * (P.hp(k, c1) - P.hp(iS, c1)) - e1 * (P.hp(k, c0) - P.hp(iS, c0))) / le);
            if (!(block_max(mx, S) > tolf)) collinear = true;
        }
    }
    if (!collinear) {
        // gift wrapping, counter-clockwise; collinear candidates: the farthest wins (interior ones dropped)
        int cur = iS;
        for (int step = 0; step < m; ++step) {
            if (tid == 0) P.setf(cur, 1);
            const double cx = P.hp(cur, c0), cy = P.hp(cur, c1);
            int best = -1; double bx = 0, by = 0;
            for (int k = tid; k < m; k += G::BT) {
                const double qx = P.hp(k, c0) - cx, qy = P.hp(k, c1) - cy;
                const double lq = qx * qx + qy * qy;
                if (!(lq > tolf * tolf)) continue;  // the current point or a duplicate of it
                if (best < 0) { best = k; bx = qx; by = qy; continue; }
                const double cr = bx * qy - by * qx, lb = bx * bx + by * by;
                // clockwise of the best so far by more than round-off, or collinear with it and farther.  Measured like
                // Qhull measures it: the nearer of the two points clears the line through the farther one by more than
                // a few ulps of the cluster's extent (Qhull merges facets flatter than that and keeps every vertex that
                // sticks out more: contact points along an edge are collinear to 1e-9 .. 1e-12 only, and whether such a
                // point survives changes the contact set)
                const double c2 = cr * cr, t2 = dtol2 * fmax(lb, lq);
                if ((cr < 0.0 && c2 > t2) || (c2 <= t2 && lq > lb)) { best = k; bx = qx; by = qy; }
            }
            S.red_i[tid] = best;
            G::sync();
            for (int s = G::BT / 2; s > 0; s >>= 1) {
                if (tid < s) {
                    const int a = S.red_i[tid], b = S.red_i[tid + s];
                    if (b >= 0) {
                        if (a < 0) S.red_i[tid] = b;
                        else {
                            const double ax = P.hp(a, c0) - cx, ay = P.hp(a, c1) - cy;
                            const double qx = P.hp(b, c0) - cx, qy = P.hp(b, c1) - cy;
                            const double cr = ax * qy - ay * qx, la = ax * ax + ay * ay, lq = qx * qx + qy * qy;
                            const double c2 = cr * cr, t2 = dtol2 * fmax(la, lq);
                            if ((cr < 0.0 && c2 > t2) || (c2 <= t2 && (lq > la || (lq == la && b < a)))) S.red_i[tid] = b;
                        }
                    }
                }
                G::sync();
            }
            const int nxt = S.red_i[0];
            const int seen = nxt >= 0 ? (P.getf(nxt) == 1) : 0;
            G::sync();   // every thread has read red_i / hflag before thread 0 flags the next vertex
            if (nxt < 0 || nxt == iS || seen) break;
            {   // coincident with the start (shared mesh vertices produce exact duplicates): the loop is closed
                const double dx = P.hp(nxt, c0) - P.hp(iS, c0), dy = P.hp(nxt, c1) - P.hp(iS, c1);
                if (!(dx * dx + dy * dy > tolf * tolf)) break;
            }
            cur = nxt;
        }
        G::sync();
        return 0;
    }
    // ---- 1-D: drop the next least-variance coordinate, keep min (and max if the spread > eps) ---
    const int keep = (var[c1] < var[c0]) ? c0 : c1;  // argmin over the remaining two drops the smaller
    double kmin = INFINITY, kmax = -INFINITY; int imin = -1, imax = -1;
    for (int k = tid; k < m; k += G::BT) {
        const double v = P.hp(k, keep);
        if (v < kmin) { kmin = v; imin = k; }
        if (v > kmax) { kmax = v; imax = k; }
    }
    const int gmin = block_argmin(kmin, imin, S), gmax = block_argmin(-kmax, imax, S);
    if (tid == 0) {
        P.setf(gmin, 1);
        if (P.hp(gmax, keep) - P.hp(gmin, keep) > eps) P.setf(gmax, 1);
    }
    G::sync();
    return 0;
}
// The escape of world.py:345-347 (strict_no_penetration=False and dt already below dt / 2^10): the step goes through with the
// contacts as _search_contacts left them when it met a penetration (contacts.py:249-272) -- every contact of this direction,
// unthinned, computed under no_grad.  They are written with NEGATIVE count: the gather stage drops the reverse direction of
// the pair (which the reference then does not search) and marks the contacts as gradient-free.
template <class G>
__device__ __noinline__ void emit_unfiltered(const DssWorld &W, int sc, int dp, int ncon, int over, const int *kface, const double *cb, int MC)
{
    const int np = npairs_of(W.nb), tid = G::tid(), MP = W.max_pc;
    int *pf = W.pc_face + ((size_t)sc * np + dp) * MP;
    double *pabc = W.pc_abc + ((size_t)sc * np + dp) * 3 * MP, *pg = W.pc_geom + ((size_t)sc * np + dp) * 10 * MP;
    int nout = ncon;
    if (nout > MP) { over |= 4; nout = MP; }
    for (int k = tid; k < nout; k += G::BT) {
        pf[k] = kface[k];
        for (int i = 0; i < 3; ++i) {
            pabc[(size_t)i * MP + k] = cb[(size_t)(15 + i) * MC + k];
            pg[(size_t)i * MP + k] = cb[(size_t)(18 + i) * MC + k];
            pg[(size_t)(3 + i) * MP + k] = cb[(size_t)(21 + i) * MC + k];
            pg[(size_t)(6 + i) * MP + k] = cb[(size_t)i * MC + k];
        }
        pg[(size_t)9 * MP + k] = cb[(size_t)24 * MC + k];
    }
    if (tid == 0) { W.pc_count[(size_t)sc * np + dp] = -nout; if (over) atomicOr(W.overflow + sc, over); }
}

// ---- stages 5 and 6 of a work item: thin the contacts (greedy normal clusters, hull of each; contacts.py:97-158) and
// write the kept ones, in ascending face order, to the pair's output slots.  On entry the candidate scratch holds, for
// contacts k < ncon: kface[k], barycentrics (fields 15-17), normal (18-20), p1 (21-23), p2 (0-2), penetration (24).
// Returns 1 if a wavefront-sized group has to hand the item to a workgroup, else 0.
#define CB(f, k) cb[(size_t)(f) * MC + (k)]
#define STAMP(i) do { if (W.dbg_stamps && G::tid() == 0) W.dbg_stamps[(size_t)item * 8 + (i)] = wall_clock64(); } while (0)
template <class G>
__device__ __forceinline__ int filter_and_emit(const DssWorld &W, ScratchT<G> &S, int item, int sc, int dp, int ncon, int over,
                                               int *__restrict__ cface, int *__restrict__ kface, int *__restrict__ cstate,
                                               double *__restrict__ cb, int MC)
{
    const int np = npairs_of(W.nb), tid = G::tid();
    int *pc_count = W.pc_count + (size_t)sc * np + dp;
#include "np_filter_emit.inc"
}
#undef STAMP
#undef CB

}  // namespace
