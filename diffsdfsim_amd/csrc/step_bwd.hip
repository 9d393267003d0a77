// step_bwd.hip -- reverse sweep over the stepper's tape (gfx950).
//
// One call of dss_step_backward undoes one accepted sub-step per scene (the newest unprocessed one):
//
//   bwd_pre_kernel    (a) adjoint of the contact geometry that was computed at the END of the
//                         sub-step (contacts.py:161-214 on the filtered set) -> pose, shape params
//                     (b) adjoint of Body3D.move (bodies.py:488-511)       -> start pose, new velocity
//                     (c) re-assembles the sub-step's LCP operands from the tape (engines.py:36-81)
//   lcp_contact_backward   implicit differentiation of the LCP (lcp.py:156-213)
//   bwd_post_kernel   adjoint of the assembly: u = M v + dt f, M = blockdiag(R I R^T, m),
//                     friction directions, mu / restitution averages, h = (Jc v) e
//                     (engines.py:36-81, physics3d/world.py:48-101, world.py:400-501)
//
// Small nonlinear stages are differentiated with forward-mode duals (geom.h) seeded a few inputs at a
// time and contracted with the incoming adjoint; linear stages are transposed by hand.  Sums over the
// contacts of a body run in contact order (no atomics): gradients are bit-reproducible.
//
// Compiled twice, like narrowphase.hip: as is (box / sphere / cylinder only) and through step_bwd_all.hip (every
// primitive; exports only launch_bwd_pre_all, which dss_step_backward uses when DssWorld.shape_rare is set).
#ifndef DSS_ALL_SHAPES
#define DSS_ALL_SHAPES 0
#endif
#include <math.h>

#include "../../include/diffsdfsim_hip.h"
#include "contact_geom.h"
#if !DSS_ALL_SHAPES
#include "contact_rev.h"
#endif
#include "wave_utils.h"

namespace {
using namespace dss;

struct SlotView {   // where the data of "sub-step k" and of "the state after it" live
    const double *pose_k, *vel_k;               // [nb][7], [nb][6] start of sub-step k
    const double *pose_n;                       // [nb][7] pose after sub-step k
    double dt;
    int nc_k; const int *body_k; const double *geom_k;                   // contacts used by the LCP of k
    int nc_n; const int *body_n, *face_n; const double *abc_n, *geom_n;  // contacts detected after k
    const double *x, *lam, *slack, *nu;
};

__device__ inline void view_slot(const DssWorld &W, int sc, int k, SlotView &v)
{
    const int nb = W.nb, MX = W.maxc, NR = W.fric_dirs + 2;
    const size_t rec = (size_t)k * W.B + sc;
    v.pose_k = W.tp_pose + rec * nb * 7;
    v.vel_k = W.tp_vel + rec * nb * 6;
    v.dt = W.tp_dt[rec];
    v.nc_k = W.tp_nc[rec];
    v.body_k = W.tp_body + rec * 2 * MX;
    v.geom_k = W.tp_geom + rec * 10 * MX;
    v.x = W.tp_x + rec * 6 * nb;
    v.lam = W.tp_lam + rec * NR * MX;
    v.slack = W.tp_slack + rec * NR * MX;
    v.nu = W.tp_nu + rec * (W.neq > 0 ? W.neq : 1);
    if (k + 1 < W.nsub[sc]) {
        const size_t r2 = (size_t)(k + 1) * W.B + sc;
        v.pose_n = W.tp_pose + r2 * nb * 7;
        v.nc_n = W.tp_nc[r2]; v.body_n = W.tp_body + r2 * 2 * MX; v.face_n = W.tp_face + r2 * MX; v.abc_n = W.tp_abc + r2 * 3 * MX;
        v.geom_n = W.tp_geom + r2 * 10 * MX;
    } else {
        v.pose_n = W.pose + (size_t)sc * nb * 7;
        v.nc_n = W.nc[sc]; v.body_n = W.c_body + (size_t)sc * 2 * MX; v.face_n = W.c_face + (size_t)sc * MX;
        v.abc_n = W.c_abc + (size_t)sc * 3 * MX;
        v.geom_n = W.c_geom + (size_t)sc * 10 * MX;
    }
}

// which sub-step a scene undoes in this call, and the view of it (shared by the kernels of one dss_step_backward)
__device__ inline void bwd_select(const DssWorld &W, const DssAdjoint &A, int sc, int &k, int &act, int &init, SlotView &v)
{
    const int nb = W.nb, MX = W.maxc;
    k = A.cur_slot[sc];
    act = (k >= 0 && k >= A.lo_slot[sc] && k < W.nsub[sc] && k < W.max_sub);   // (a slot beyond the tape was never recorded)
    // slot -1 = the contacts found at construction (World.__init__, world.py:96): only their geometry
    // adjoint is left to push onto the initial pose and the shape parameters
    init = (k == -1 && A.lo_slot[sc] <= -1 && W.nsub[sc] > 0);
    if (init) {
        const size_t r0 = (size_t)sc;   // tape slot 0
        v.pose_n = W.tp_pose + r0 * nb * 7;
        v.nc_n = W.tp_nc[r0]; v.body_n = W.tp_body + r0 * 2 * MX; v.face_n = W.tp_face + r0 * MX; v.abc_n = W.tp_abc + r0 * 3 * MX;
        v.geom_n = W.tp_geom + r0 * 10 * MX;
    } else if (act) {
        view_slot(W, sc, k, v);
    }
}

// d(n, p1, p2)/d(pose1, pose2, prm1, prm2) contracted with gbar[9]; out[20] = q1(4) x1(3) q2(4) x2(3) prm1(3) prm2(3).
// Forward-mode duals, four seeds per pass.  Only q1 and prm1 enter the body-1 half of the contact (contact_head: two
// SDF queries and the Newton step): they take two full passes.  q2, x2 and prm2 enter the body-2 half alone
// (contact_tail: one query, two rotations), so their three passes differentiate that half with the head as constants;
// x1 appears only in rel = p1 + x1 - x2, hence d/dx1 = -d/dx2 and needs no pass of its own.
#if DSS_ALL_SHAPES
template <class T> __device__ inline void attach_grid(const DssWorld &W, int sc, int b, Shape<T> &s)
{
    if (s.type != SHAPE_GRID || !W.grid_id) return;
    const int gi = W.grid_id[(size_t)sc * W.nb + b];
    s.grid = W.grid_data + W.grid_off[gi];
    for (int i = 0; i < 3; ++i) s.gn[i] = W.grid_dims[3 * gi + i];
}
#endif
// lin1 / lin2: igr_lin records of a neural body 1 / body 2 for this contact (NULL: analytic body); stable_in >= 0: which
// body's normal the contact used, decided by the caller (for neural bodies the Laplacian probes are not repeated).
// Forward mode (five dual-number passes): the full variant (every primitive, neural / grid bodies, mesh-vertex adjoints).  The
// lean variant -- box / sphere / cylinder, what the benchmark configs run -- uses the reverse-mode adjoint of contact_rev.h.
#if DSS_ALL_SHAPES || defined(DSS_BWD_FORWARD_MODE)
__device__ void contact_vjp(const DssWorld &W, int sc, const double *pose_n, int b1, int b2, int face,
                            const double *abc, const double *gbar, double *out, double *g_verts,
                            const double *lin1 = nullptr, const double *lin2 = nullptr, int stable_in = -1)
{
    constexpr int N = 4;
    typedef Dual<N> D;
    const int nb = W.nb;
    const bool det2 = (W.grad_flags & DSS_GRAD_DETACH_B2) != 0;
    const double *P1 = pose_n + 7 * b1, *P2 = pose_n + 7 * b2;
    const double *prm1 = W.shape_prm + ((size_t)sc * nb + b1) * 3, *prm2 = W.shape_prm + ((size_t)sc * nb + b2) * 3;
    const int ty1 = W.shape_type[(size_t)sc * nb + b1], ty2 = W.shape_type[(size_t)sc * nb + b2];
#if DSS_ALL_SHAPES
    const double aux1 = W.shape_aux[(size_t)sc * nb + b1], aux2 = W.shape_aux[(size_t)sc * nb + b2];
#else
    const double aux1 = 0.0, aux2 = 0.0;
#endif
    const int mesh = W.mesh_id[(size_t)sc * nb + b1];
    const int voff = W.mesh_voff[mesh], foff = W.mesh_foff[mesh];
    const int *fv = W.faces + (size_t)(foff + face) * 3;
    double tv[3][3], tg[3][3];
    for (int v = 0; v < 3; ++v)
        for (int i = 0; i < 3; ++i) { tv[v][i] = W.verts[(size_t)(voff + fv[v]) * 3 + i]; tg[v][i] = W.vgrad[(size_t)(voff + fv[v]) * 3 + i]; }
    // value pass: the head as constants for the body-2 passes, and the normal-selection decision for all of them
    int stable = stable_in;
    double cp1v[3], n1v[3], d1v, p1v[3];
    {
        BodyG<double> B1, B2;
        for (int i = 0; i < 4; ++i) { B1.q[i] = P1[i]; B2.q[i] = P2[i]; }
        for (int i = 0; i < 3; ++i) { B1.pos[i] = P1[4 + i]; B2.pos[i] = P2[4 + i]; }
        make_shape(B1.shape, ty1, prm1, aux1);
        make_shape(B2.shape, ty2, prm2, aux2);
#if DSS_ALL_SHAPES
        B1.shape.lin = lin1; B2.shape.lin = lin2; attach_grid(W, sc, b1, B1.shape); attach_grid(W, sc, b2, B2.shape);
#endif
        double nn[3], pp2[3], pen;
        contact_head(B1, tv, abc, cp1v, n1v, d1v, p1v);
        contact_tail(B1, B2, cp1v, n1v, d1v, p1v, 1e-3, nn, pp2, pen, &stable);
    }
    for (int t = 0; t < 20; ++t) out[t] = 0.0;
    auto contract = [&](const D *n, const D *p1, const D *p2, int slot0, int cnt) {
        for (int s = 0; s < cnt; ++s) {
            double acc = 0.0;
            for (int i = 0; i < 3; ++i) acc += gbar[i] * n[i].d[s] + (p1 ? gbar[3 + i] * p1[i].d[s] : 0.0) + gbar[6 + i] * p2[i].d[s];
            out[slot0 + s] = acc;
        }
    };
    // ---- body-1 inputs: full passes, seeds q1 | prm1 -------------------------------------------------------
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
        BodyG<D> B1, B2;
        D pr1[3], pr2[3];
        for (int i = 0; i < 4; ++i) { B1.q[i] = D(P1[i]); if (grp == 0) B1.q[i].d[i] = 1.0; B2.q[i] = D(P2[i]); }
        for (int i = 0; i < 3; ++i) {
            B1.pos[i] = D(P1[4 + i]); B2.pos[i] = D(P2[4 + i]);
            pr1[i] = D(prm1[i]); if (grp == 1) pr1[i].d[i] = 1.0;
            pr2[i] = D(prm2[i]);
        }
        make_shape(B1.shape, ty1, pr1, aux1);
        make_shape(B2.shape, ty2, pr2, aux2);
#if DSS_ALL_SHAPES
        B1.shape.lin = lin1; B2.shape.lin = lin2; attach_grid(W, sc, b1, B1.shape); attach_grid(W, sc, b2, B2.shape);
#endif
        D tri[3][3];
        for (int v = 0; v < 3; ++v)
            for (int i = 0; i < 3; ++i) {
                D d(tv[v][i]);
                if (grp == 1) {
                    // box: own axis; sphere: radius; cylinder: x,y <- rad, z <- height
                    const int s = (ty1 == SHAPE_BOX || ty1 == SHAPE_BOX_ROUNDED || ty1 == SHAPE_BRICK) ? i : ((ty1 == SHAPE_CYLINDER && i == 2) ? 1 : 0);
#pragma unroll
                    for (int sl = 0; sl < 3; ++sl) if (sl == s) d.d[sl] = tg[v][i];   // selects: a run-time index would put d in scratch
                }
                tri[v][i] = d;
            }
        D n[3], p1[3], p2[3], pen;
        contact_from_bary(B1, B2, tri, abc, 1e-3, n, p1, p2, pen, &stable, det2);
        if (grp == 0) contract(n, p1, p2, 0, 4);
        else contract(n, p1, p2, 14, 3);
    }
    // ---- body-2 inputs: the tail alone, seeds q2 | x2 | prm2 -----------------------------------------------
#pragma unroll
    for (int grp = 0; grp < 3; ++grp) {
        BodyG<D> B1, B2;
        D pr1[3], pr2[3];
        for (int i = 0; i < 4; ++i) { B1.q[i] = D(P1[i]); B2.q[i] = D(P2[i]); if (grp == 0) B2.q[i].d[i] = 1.0; }
        for (int i = 0; i < 3; ++i) {
            B1.pos[i] = D(P1[4 + i]);
            B2.pos[i] = D(P2[4 + i]); if (grp == 1) B2.pos[i].d[i] = 1.0;
            pr1[i] = D(prm1[i]);
            pr2[i] = D(prm2[i]); if (grp == 2) pr2[i].d[i] = 1.0;
        }
        make_shape(B1.shape, ty1, pr1, aux1);
        make_shape(B2.shape, ty2, pr2, aux2);
#if DSS_ALL_SHAPES
        B1.shape.lin = lin1; B2.shape.lin = lin2; attach_grid(W, sc, b1, B1.shape); attach_grid(W, sc, b2, B2.shape);
#endif
        D cp1[3], n1[3], d1(d1v), p1[3], n[3], p2[3], pen;
        for (int i = 0; i < 3; ++i) { cp1[i] = D(cp1v[i]); n1[i] = D(n1v[i]); p1[i] = D(p1v[i]); }
        contact_tail(B1, B2, cp1, n1, d1, p1, 1e-3, n, p2, pen, &stable, det2);
        if (grp == 0) contract(n, nullptr, p2, 7, 4);
        else if (grp == 1) { contract(n, nullptr, p2, 11, 3); for (int i = 0; i < 3; ++i) out[4 + i] = -out[11 + i]; }
        else contract(n, nullptr, p2, 17, 3);
    }
#if DSS_ALL_SHAPES
    // ---- the triangle's vertices (full variant): one full pass per vertex, seeds = its three coordinates -----------
    // Level-set meshes have no per-vertex parameter tangent (vgrad = 0); their shape gradient flows through the vertex
    // positions themselves and is chained to the parameters by the mesher's backward (MeshSDF, bodies.py:680-702).
    if (g_verts) {
        for (int vtx = 0; vtx < 3; ++vtx) {
            BodyG<D> B1, B2;
            D pr1[3], pr2[3];
            for (int i = 0; i < 4; ++i) { B1.q[i] = D(P1[i]); B2.q[i] = D(P2[i]); }
            for (int i = 0; i < 3; ++i) { B1.pos[i] = D(P1[4 + i]); B2.pos[i] = D(P2[4 + i]); pr1[i] = D(prm1[i]); pr2[i] = D(prm2[i]); }
            make_shape(B1.shape, ty1, pr1, aux1);
            make_shape(B2.shape, ty2, pr2, aux2);
#if DSS_ALL_SHAPES
            B1.shape.lin = lin1; B2.shape.lin = lin2; attach_grid(W, sc, b1, B1.shape); attach_grid(W, sc, b2, B2.shape);
#endif
            D tri[3][3];
            for (int v = 0; v < 3; ++v)
                for (int i = 0; i < 3; ++i) {
                    D d(tv[v][i]);
#pragma unroll
                    for (int sl = 0; sl < 3; ++sl) if (v == vtx && sl == i) d.d[sl] = 1.0;
                    tri[v][i] = d;
                }
            D n[3], p1[3], p2[3], pen;
            contact_from_bary(B1, B2, tri, abc, 1e-3, n, p1, p2, pen, &stable, det2);
            for (int s = 0; s < 3; ++s) {
                double acc = 0.0;
                for (int i = 0; i < 3; ++i) acc += gbar[i] * n[i].d[s] + gbar[3 + i] * p1[i].d[s] + gbar[6 + i] * p2[i].d[s];
                atomicAdd(g_verts + (size_t)(voff + fv[vtx]) * 3 + s, acc);
            }
        }
    }
#endif
}

#endif   // forward-mode contact_vjp

#if DSS_ALL_SHAPES
// ---- neural SDF bodies in the reverse sweep ------------------------------------------------------------------------------
// The contact geometry of a neural body is differentiated through records of the network at the points it was queried at
// (geom.h: igr_lin).  bwd_igr_prep_kernel lists those points for the sub-step every scene is about to undo -- body 1 at the
// barycentric point of the contact's triangle, body 2 at the contact point in its frame -- igr_query_kernel evaluates the
// list twice on the matrix cores (d/dxyz, d/dlatent), igr_records turns the answers into the records of one contact.
__device__ inline bool in_cube3(const double *p, double s) { return fabs(p[0]) <= s && fabs(p[1]) <= s && fabs(p[2]) <= s; }

__global__ void __launch_bounds__(64) bwd_igr_prep_kernel(DssWorld W_arg, DssAdjoint A_arg)
{
    static_assert(sizeof(DssWorld) % 8 == 0, "the adjoint descriptor follows the world descriptor without padding");
    DSS_KERNARG_REF(DssWorld, W, W_arg);
    DSS_KERNARG_REF_AT(DssAdjoint, A, A_arg, sizeof(DssWorld));
    const int sc = blockIdx.x, lane = threadIdx.x, nb = W.nb, MX = W.maxc;
    int k, act, init;
    SlotView v;
    bwd_select(W, A, sc, k, act, init, v);
    if (!act && !init) return;
    for (int c = lane; c < v.nc_n; c += 64) {
        const int b1 = v.body_n[c], b2 = v.body_n[MX + c];
        int idx[2] = {-1, -1};
        for (int side = 0; side < 2 && v.face_n[c] >= 0; ++side) {
            const int b = side ? b2 : b1;
            if (W.shape_type[(size_t)sc * nb + b] != DSS_SHAPE_IGR) continue;
            const double scale = W.shape_aux[(size_t)sc * nb + b];
            double pt[3];
            if (side == 0) {
                const int mesh = W.mesh_id[(size_t)sc * nb + b1];
                const int *fv = W.faces + (size_t)(W.mesh_foff[mesh] + DSS_FACE_ID(v.face_n[c])) * 3;
                pt[0] = pt[1] = pt[2] = 0.0;
                for (int q = 0; q < 3; ++q) {
                    const double *vp = W.verts + (size_t)(W.mesh_voff[mesh] + fv[q]) * 3, w = v.abc_n[(size_t)q * MX + c];
                    for (int i = 0; i < 3; ++i) pt[i] = pt[i] + vp[i] * w;        // (the order of contact_head)
                }
                // contact_head forms tri[0] abc[0] + tri[1] abc[1] + tri[2] abc[2]: the same sum, left to right
            } else {
                const double *P1 = v.pose_n + 7 * b1, *P2 = v.pose_n + 7 * b2;
                double rel[3];
                for (int i = 0; i < 3; ++i) rel[i] = (v.geom_n[(size_t)(3 + i) * MX + c] + P1[4 + i]) - P2[4 + i];
                quat_apply_inv(P2, rel, pt);
            }
            if (!in_cube3(pt, scale)) { idx[side] = -2; continue; }       // query_sdfs: phi = scale, grad = 0 out there
            const int slot = atomicAdd(A.igr_bw_n, 1);
            double u[3];
            div3(pt, scale, u);
            for (int i = 0; i < 3; ++i) A.igr_bw_pts[(size_t)slot * 3 + i] = u[i];
            A.igr_bw_lat[slot] = sc * nb + b;
            idx[side] = slot;
        }
        A.igr_bw_idx[((size_t)sc * 2 + 0) * MX + c] = idx[0];
        A.igr_bw_idx[((size_t)sc * 2 + 1) * MX + c] = idx[1];
    }
}

// records of contact c: lin[0 .. 2 IGR_LIN) body 1 (queries at the triangle point and after the Newton step), lin[2 IGR_LIN ..)
// body 2; `stable` = which normal the forward pass picked (the flag in the contact's face word)
__device__ void igr_records(const DssWorld &W, const DssAdjoint &A, int sc, const SlotView &v, int c, int i1, int i2, double *lin,
                            int &stable)
{
    const int nb = W.nb, MX = W.maxc, b1 = v.body_n[c], b2 = v.body_n[MX + c];
    const size_t cap = (size_t)W.B * 2 * MX;
    const double *sdfX = A.igr_bw_sdf, *gX = A.igr_bw_grad, *gL = A.igr_bw_grad + cap * 3;
    const double *P1 = v.pose_n + 7 * b1;
    for (int i = 0; i < 3 * IGR_LIN; ++i) lin[i] = 0.0;
    auto fill = [&](double *r, int idx, double scale) {
        if (idx < 0) { r[0] = scale; return; }      // outside the query cube: phi = scale, everything else zero
        r[0] = sdfX[idx] * scale;
        const double raw[3] = {gX[(size_t)idx * 3], gX[(size_t)idx * 3 + 1], gX[(size_t)idx * 3 + 2]};
        for (int i = 0; i < 3; ++i) r[1 + i] = raw[i];                  // d (scale f(pt / scale)) / d pt
        for (int j = 0; j < 2; ++j) r[4 + j] = gL[(size_t)idx * 3 + j] * scale;
        normalize(raw, r + 6);
    };
    if (i1 != -1) fill(lin, i1, W.shape_aux[(size_t)sc * nb + b1]);
    if (i2 != -1) fill(lin + 2 * IGR_LIN, i2, W.shape_aux[(size_t)sc * nb + b2]);
    // which body's normal the contact carries was decided in the forward pass and travels with the face id
    const double nt[3] = {v.geom_n[c], v.geom_n[(size_t)MX + c], v.geom_n[(size_t)2 * MX + c]};
    stable = (v.face_n[c] & DSS_FACE_NORMAL1) ? 0 : 1;
    if (i1 != -1 && !stable) {       // the normal used is body 1's after the Newton step: n = -R1 n1'  ->  n1' = -R1^T n
        double t[3];
        quat_apply_inv(P1, nt, t);
        for (int i = 0; i < 3; ++i) lin[IGR_LIN + 6 + i] = -t[i];
    }
}
#endif

// Ordered per-body sums of per-contact pieces:
//   sums[b * NC + q] = sum over contacts c, in contact order, of [body1(c) = b] cs[row0[q]][c] + [body2(c) = b] cs[row1[q]][c].
// One lane per (body, component) instead of one lane per body walking every component: the loads of a contact do not
// depend on the running sums, so the unrolled loop keeps four contacts in flight; a term that does not belong to the
// lane's body is added as 0.0, which leaves the sum -- and therefore its summation order -- exactly as before.
template <int NC>
__device__ inline void contact_sums(const int *body, int MX, int nc, int nb, const double *cs, const int *row0,
                                    const int *row1, double *sums)
{
    const int lane = threadIdx.x;
    for (int e = lane; e < nb * NC; e += 64) {
        const int b = e / NC, q = e % NC;
        const double *c0 = cs + (size_t)row0[q] * MX, *c1 = cs + (size_t)row1[q] * MX;
        double acc = 0.0;
        int c = 0;
        for (; c + 4 <= nc; c += 4) {
            int b1[4], b2[4];
            double v0[4], v1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { b1[u] = body[c + u]; b2[u] = body[MX + c + u]; v0[u] = c0[c + u]; v1[u] = c1[c + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc += (b1[u] == b) ? v0[u] : 0.0; acc += (b2[u] == b) ? v1[u] : 0.0; }
        }
        for (; c < nc; ++c) { acc += (body[c] == b) ? c0[c] : 0.0; acc += (body[MX + c] == b) ? c1[c] : 0.0; }
        sums[e] = acc;
    }
    __syncthreads();
}

__global__ void __launch_bounds__(64) bwd_pre_kernel(DssWorld W_arg, DssAdjoint A_arg)
{
    static_assert(sizeof(DssWorld) % 8 == 0, "the adjoint descriptor follows the world descriptor without padding");
    DSS_KERNARG_REF(DssWorld, W, W_arg);
    DSS_KERNARG_REF_AT(DssAdjoint, A, A_arg, sizeof(DssWorld));
    const int sc = blockIdx.x, lane = threadIdx.x, nb = W.nb, MX = W.maxc;
    int k, act, init;
    SlotView v;
    bwd_select(W, A, sc, k, act, init, v);
    if (lane == 0) A.bw_active[sc] = act;
    if (!act && !init) return;
    double *a_pose = A.a_pose + (size_t)sc * nb * 7, *a_vel = A.a_vel + (size_t)sc * nb * 6;
    double *a_geom = A.a_geom + (size_t)sc * 10 * MX, *cs = A.cscr + (size_t)sc * DSS_CSCR_ROWS * MX;

    // (0) time-of-contact event (world.py:272-341): dt_h = H(dt_, theta).  Its adjoint is that of the redone
    //     move plus the carry from the next sub-step (whose dt_ = -last_dt + ...); H.backward (world.py:195-237)
    //     turns it into adjoints of the new contacts' geometry, the new velocities, the moved poses and f/m.
    const int flags = init ? 0 : W.tp_flags[(size_t)k * W.B + sc];
    const int ev = flags & 1;
    double dt_int = 0.0;   // d(loss)/d(dt) through the pose integration, seed = pose adjoint before the TOC terms
    // Jacobian of this body's move (Body3D.move, bodies.py:488-496: q' = quat(exp(w dt)) (x) q, x' = x + v dt) from ONE
    // dual-number pass over theta = w dt: d q'/d theta (4 x 3).  Everything else follows in closed form -- d/dw = dt d/dtheta,
    // d/d dt = sum_i w_i d/dtheta_i, q' is linear in q (adjoint = conj(dq) (x) .), x' is affine -- where four full passes
    // (one seed, seven, six and one) used to set this kernel's register footprint.
    double Jt[4][3], dqv[4] = {1.0, 0.0, 0.0, 0.0}, qsg = 1.0, vnw[6] = {0, 0, 0, 0, 0, 0};
    for (int o = 0; o < 4; ++o) for (int s3 = 0; s3 < 3; ++s3) Jt[o][s3] = 0.0;
    if (!init && lane < nb) {
        typedef Dual<3> D;
        for (int i = 0; i < 6; ++i) vnw[i] = -v.x[6 * lane + i];
        D w[3], R[9], dq[4], qk[4], o4[4];
        for (int i = 0; i < 3; ++i) { w[i] = D(vnw[i] * v.dt); w[i].d[i] = 1.0; }
        so3_exp(w, R);
        mat_to_quat(R, dq);
        for (int i = 0; i < 4; ++i) qk[i] = D(v.pose_k[7 * lane + i]);
        quat_raw_mul(dq, qk, o4);
        qsg = o4[0].v < 0.0 ? -1.0 : 1.0;      // quaternion_multiply standardises to a non-negative real part
        for (int o = 0; o < 4; ++o) { dqv[o] = dq[o].v; for (int s3 = 0; s3 < 3; ++s3) Jt[o][s3] = qsg * o4[o].d[s3]; }
    }
    // adjoint of the move for a pose adjoint ap[7]: -> pose_k (apk), v_new (avn), dt (returned)
    auto move_adjoint = [&](const double *ap, double *apk, double *avn) -> double {
        double th[3], adt = 0.0;
        for (int s3 = 0; s3 < 3; ++s3) th[s3] = ap[0] * Jt[0][s3] + ap[1] * Jt[1][s3] + ap[2] * Jt[2][s3] + ap[3] * Jt[3][s3];
        for (int i = 0; i < 3; ++i) { adt += th[i] * vnw[i] + ap[4 + i] * vnw[3 + i]; if (avn) { avn[i] = th[i] * v.dt; avn[3 + i] = ap[4 + i] * v.dt; } }
        if (apk) {
            const double dc[4] = {dqv[0], -dqv[1], -dqv[2], -dqv[3]}, aq[4] = {qsg * ap[0], qsg * ap[1], qsg * ap[2], qsg * ap[3]};
            quat_raw_mul(dc, aq, apk);          // <a, dq (x) q> = <conj(dq) (x) a, q>
            for (int i = 0; i < 3; ++i) apk[4 + i] = ap[4 + i];
        }
        return adt;
    };
    if (!init) {
        double part = 0.0;
        if (lane < nb) part = move_adjoint(a_pose + 7 * lane, nullptr, nullptr);
        dt_int = wave_sum(part);
    }
    double hc_bar = 0.0;
    if (ev) {
        const double dtbar_h = dt_int + A.a_last_dt[sc];
        const double h = v.dt;
        double dDdh[3] = {0, 0, 0};
        int q = 0;
        double den = 0.0;
        auto fill = [&](int c, double *in) {
            const int b1 = v.body_n[c], b2 = v.body_n[MX + c];
            in[0] = h; in[1] = h;
            const double *gn = (k + 1 < W.nsub[sc]) ? W.tp_geom + ((size_t)(k + 1) * W.B + sc) * 10 * MX : W.c_geom + (size_t)sc * 10 * MX;
            for (int i = 0; i < 3; ++i) { in[2 + i] = gn[(size_t)(3 + i) * MX + c]; in[5 + i] = gn[(size_t)(6 + i) * MX + c]; in[8 + i] = gn[(size_t)i * MX + c]; }
            for (int i = 0; i < 6; ++i) { in[11 + i] = -v.x[6 * b1 + i]; in[17 + i] = -v.x[6 * b2 + i]; }
            for (int i = 0; i < 7; ++i) { in[23 + i] = v.pose_n[7 * b1 + i]; in[30 + i] = v.pose_n[7 * b2 + i]; }
            for (int i = 0; i < 3; ++i) {
                in[37 + i] = W.fext[((size_t)sc * nb + b1) * 6 + 3 + i] / W.mass[(size_t)sc * nb + b1];
                in[40 + i] = W.fext[((size_t)sc * nb + b2) * 6 + 3 + i] / W.mass[(size_t)sc * nb + b2];
            }
        };
        auto is_toc = [&](int c) {
            const int a = v.body_n[c], b = v.body_n[MX + c];
            for (int j = 0; j < v.nc_k; ++j) {
                const int a0 = v.body_k[j], b0 = v.body_k[MX + j];
                if ((a0 == a && b0 == b) || (a0 == b && b0 == a)) return false;
            }
            return true;
        };
        auto dD_dh = [&](int c) -> double {
            if (!is_toc(c)) return 0.0;
            double in[43];
            fill(c, in);
            typedef Dual<1> D;
            D di[43];
            for (int i = 0; i < 43; ++i) di[i] = D(in[i]);
            di[0].d[0] = 1.0;
            const double g = toc_D(di).d[0];
            return g < 1e-6 / h ? 0.0 : g;      // only motion into collision (world.py:203; Defaults.TOL = 1e-6)
        };
        for (int c = lane; c < v.nc_n; c += 64, ++q) {
            const double g = dD_dh(c);
            if (q < 3) dDdh[q] = g;             // the first three contacts of a lane are cached, later ones recomputed below
            den += g * g;
        }
        den = wave_sum(den);
        q = 0;
        for (int c = lane; c < v.nc_n; c += 64) for (int r = 20; r < 53; ++r) cs[(size_t)r * MX + c] = 0.0;
        __syncthreads();
        // The gradient of D with respect to its 43 inputs, one dual-number pass per input.  A time-of-contact event is rare (a
        // handful per rollout and scene) but its sweep iteration is the slowest scene's: with the 43 passes walked by the lane
        // that owns the contact, a batch of free-running scenes -- where some scene meets an event in almost every iteration --
        // spent 270 us per iteration here.  The passes of ONE contact go to 43 lanes instead (lane = seed); the contacts of an
        // event (one to four) are taken one after the other.
        for (int base = 0; base < v.nc_n; base += 64, ++q) {
            const int c_own = base + lane;
            double gq = 0.0;
            if (c_own < v.nc_n && den > 1e-5) gq = q < 3 ? dDdh[q] : dD_dh(c_own);
            unsigned long long todo = __ballot(gq != 0.0);
            while (todo) {
                const int bit = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int c = base + bit;
                const double wgt = -(__shfl(gq, bit, 64) / den) * dtbar_h;
                double in[43];
                fill(c, in);
                double og = 0.0;
                if (lane < 43) {
                    typedef Dual<1> D;
                    D di[43];
                    for (int i = 0; i < 43; ++i) { di[i] = D(in[i]); di[i].d[0] = (i == lane) ? 1.0 : 0.0; }
                    og = wgt * toc_D(di).d[0];
                }
                // lane sd holds d/d in[sd]: geometry of the new contact (p1: 2-4, p2: 5-7, normal: 8-10), the two new velocities
                // (11-16, 17-22), the two moved poses (23-29, 30-36), f/m of both bodies (37-39, 40-42), h (1)
                const int sd = lane;
                if (sd >= 2 && sd <= 4) a_geom[(size_t)(3 + sd - 2) * MX + c] += og;
                else if (sd >= 5 && sd <= 7) a_geom[(size_t)(6 + sd - 5) * MX + c] += og;
                else if (sd >= 8 && sd <= 10) a_geom[(size_t)(sd - 8) * MX + c] += og;
                else if (sd >= 23 && sd <= 29) cs[(size_t)(20 + sd - 23) * MX + c] = og;
                else if (sd >= 30 && sd <= 36) cs[(size_t)(27 + sd - 30) * MX + c] = og;
                else if (sd >= 11 && sd <= 16) cs[(size_t)(34 + sd - 11) * MX + c] = og;
                else if (sd >= 17 && sd <= 22) cs[(size_t)(40 + sd - 17) * MX + c] = og;
                else if (sd >= 37 && sd <= 39) cs[(size_t)(46 + sd - 37) * MX + c] = og;
                else if (sd >= 40 && sd <= 42) cs[(size_t)(49 + sd - 40) * MX + c] = og;
                else if (sd == 1) cs[(size_t)52 * MX + c] = og;
            }
        }
        __syncthreads();
        double hp = 0.0;
        for (int c = lane; c < v.nc_n; c += 64) hp += cs[(size_t)52 * MX + c];
        hc_bar = wave_sum(hp);
    }

    // (a) contacts detected after the sub-step: geometry adjoint -> pose after the sub-step, shape params
    for (int c = lane; c < v.nc_n; c += 64) {
        double gb[9], out[20];
        for (int i = 0; i < 9; ++i) gb[i] = a_geom[(size_t)i * MX + c];
        const double abc[3] = {v.abc_n[c], v.abc_n[MX + c], v.abc_n[2 * MX + c]};
        if (v.face_n[c] < 0) {      // a contact kept from a penetrating direction (world.py:345-347): computed under no_grad
            for (int i = 0; i < 20; ++i) cs[(size_t)i * MX + c] = 0.0;
            continue;
        }
        const double *l1 = nullptr, *l2 = nullptr;
        int st = -1;
#if DSS_ALL_SHAPES
        double lin[3 * IGR_LIN];
        if (A.igr_bw_idx) {
            const int i1 = A.igr_bw_idx[((size_t)sc * 2 + 0) * MX + c], i2 = A.igr_bw_idx[((size_t)sc * 2 + 1) * MX + c];
            if (i1 != -1 || i2 != -1) {
                igr_records(W, A, sc, v, c, i1, i2, lin, st);
                if (i1 != -1) l1 = lin;
                if (i2 != -1) l2 = lin + 2 * IGR_LIN;
            }
        }
#endif
        // the forward pass's normal choice travels with the face id: the Laplacian probes are not repeated
        if (st < 0) st = (v.face_n[c] & DSS_FACE_NORMAL1) ? 0 : 1;
#if DSS_ALL_SHAPES || defined(DSS_BWD_FORWARD_MODE)
        contact_vjp(W, sc, v.pose_n, v.body_n[c], v.body_n[MX + c], DSS_FACE_ID(v.face_n[c]), abc, gb, out, A.g_verts, l1, l2, st);
#else
        {   // reverse mode (contact_rev.h): one value pass, one adjoint pass
            const int b1 = v.body_n[c], b2 = v.body_n[MX + c];
            const size_t i1 = (size_t)sc * nb + b1, i2 = (size_t)sc * nb + b2;
            const int mesh = W.mesh_id[i1];
            const int voff = W.mesh_voff[mesh];
            const int *fv = W.faces + (size_t)(W.mesh_foff[mesh] + DSS_FACE_ID(v.face_n[c])) * 3;
            double tv[3][3], tg[3][3];
            for (int vv = 0; vv < 3; ++vv)
                for (int i = 0; i < 3; ++i) { tv[vv][i] = W.verts[(size_t)(voff + fv[vv]) * 3 + i]; tg[vv][i] = W.vgrad[(size_t)(voff + fv[vv]) * 3 + i]; }
            contact_vjp_rev(v.pose_n + 7 * b1, v.pose_n + 7 * b2, W.shape_type[i1], W.shape_type[i2], W.shape_prm + i1 * 3, W.shape_prm + i2 * 3,
                            tv, tg, abc, gb, st, (W.grad_flags & DSS_GRAD_DETACH_B2) != 0, out);
        }
#endif
        for (int i = 0; i < 20; ++i) cs[(size_t)i * MX + c] = out[i];
    }
    __syncthreads();
    __shared__ double s_sums[64 * 10];
    {
        static constexpr int row0[10] = {0, 1, 2, 3, 4, 5, 6, 14, 15, 16}, row1[10] = {7, 8, 9, 10, 11, 12, 13, 17, 18, 19};
        contact_sums<10>(v.body_n, MX, v.nc_n, nb, cs, row0, row1, s_sums);
    }
    if (lane < nb) {
        double ap[7], gp[3];
        for (int i = 0; i < 7; ++i) ap[i] = a_pose[7 * lane + i] + s_sums[10 * lane + i];
        for (int i = 0; i < 3; ++i) gp[i] = s_sums[10 * lane + 7 + i];
        for (int i = 0; i < 3; ++i) A.g_prm[((size_t)sc * nb + lane) * 3 + i] += gp[i];
        if (ev) {   // pieces of H.backward that land on this body: moved pose, new velocity, f/m
            double vx[6] = {0, 0, 0, 0, 0, 0}, ab[3] = {0, 0, 0};
            for (int c = 0; c < v.nc_n; ++c) {
                if (v.body_n[c] == lane) {
                    for (int i = 0; i < 7; ++i) ap[i] += cs[(size_t)(20 + i) * MX + c];
                    for (int i = 0; i < 6; ++i) vx[i] += cs[(size_t)(34 + i) * MX + c];
                    for (int i = 0; i < 3; ++i) ab[i] += cs[(size_t)(46 + i) * MX + c];
                }
                if (v.body_n[MX + c] == lane) {
                    for (int i = 0; i < 7; ++i) ap[i] += cs[(size_t)(27 + i) * MX + c];
                    for (int i = 0; i < 6; ++i) vx[i] += cs[(size_t)(40 + i) * MX + c];
                    for (int i = 0; i < 3; ++i) ab[i] += cs[(size_t)(49 + i) * MX + c];
                }
            }
            const size_t bi = (size_t)sc * nb + lane;
            const double m = W.mass[bi];
            for (int i = 0; i < 3; ++i) {   // a = f/m
                A.g_fext[bi * 6 + 3 + i] += ab[i] / m;
                A.g_mass[bi] -= ab[i] * W.fext[bi * 6 + 3 + i] / (m * m);
            }
            for (int i = 0; i < 6; ++i) a_vel[6 * lane + i] += vx[i];   // joins the adjoint of the new velocity
        }
        for (int i = 0; i < 7; ++i) a_pose[7 * lane + i] = ap[i];
    }
    if (init) {
        __syncthreads();
        for (int c = lane; c < MX; c += 64) for (int i = 0; i < 10; ++i) a_geom[(size_t)i * MX + c] = 0.0;
        if (lane == 0) A.cur_slot[sc] = -2;
        return;
    }
    if (lane < nb) {
        double ap[7];
        for (int i = 0; i < 7; ++i) ap[i] = a_pose[7 * lane + i] ;
        // (b) pose_n = integrate(pose_k, v_new, dt): adjoint -> pose_k, v_new, dt (with the complete pose adjoint: first move
        //     and redone move share it)
        double apk[7], avn[6];
        cs[(size_t)53 * MX + lane] = move_adjoint(ap, apk, avn);
        for (int i = 0; i < 7; ++i) a_pose[7 * lane + i] = apk[i];
        // total adjoint of v_new = (later uses, already in a_vel) + (integration); x = -v_new
        for (int i = 0; i < 6; ++i) A.a_x[(size_t)sc * 6 * nb + 6 * lane + i] = -(a_vel[6 * lane + i] + avn[i]);
        // (c) LCP operands of sub-step k: mass blocks
        const size_t bi = (size_t)sc * nb + lane;
        double Iw[9];
        world_inertia(v.pose_k + 7 * lane, W.inertia + bi * 9, Iw);
        double *M = W.Mblk + bi * 36;
        const double m = W.mass[bi];
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) M[6 * r + c] = (r < 3 && c < 3) ? Iw[3 * r + c] : ((r == c) ? m : 0.0);
        for (int i = 0; i < 6; ++i) W.x[(size_t)sc * 6 * nb + 6 * lane + i] = v.x[6 * lane + i];
    }
    __syncthreads();
    {
        double part = (lane < nb) ? cs[(size_t)53 * MX + lane] : 0.0;
        part = wave_sum(part);
        if (lane == 0) A.a_dt[sc] = part + hc_bar + (ev ? A.a_last_dt[sc] : 0.0);
    }
    if (lane < W.neq) W.nu[(size_t)sc * W.neq + lane] = v.nu[lane];
    const int ND = W.fric_dirs / 2, NF = 3 * (1 + ND) + 8, NR = W.fric_dirs + 2;
    double *cop = W.cop + (size_t)sc * NF * MX;
    for (int c = lane; c < v.nc_k; c += 64) {
        const int b1 = v.body_k[c], b2 = v.body_k[MX + c];
        W.cop_body[(size_t)sc * 2 * MX + c] = b1;
        W.cop_body[(size_t)sc * 2 * MX + MX + c] = b2;
        double n[3], p1[3], p2[3], D[4][3];
        for (int i = 0; i < 3; ++i) { n[i] = v.geom_k[(size_t)i * MX + c]; p1[i] = v.geom_k[(size_t)(3 + i) * MX + c]; p2[i] = v.geom_k[(size_t)(6 + i) * MX + c]; }
        friction_dirs(n, ND, D);
        for (int i = 0; i < 3; ++i) {
            cop[(size_t)i * MX + c] = n[i];
            for (int q = 0; q < ND; ++q) cop[(size_t)(3 * (q + 1) + i) * MX + c] = D[q][i];
        }
        const int o = 3 * (1 + ND);
        for (int i = 0; i < 3; ++i) { cop[(size_t)(o + i) * MX + c] = p1[i]; cop[(size_t)(o + 3 + i) * MX + c] = p2[i]; }
        cop[(size_t)(o + 6) * MX + c] = 0.5 * (W.fric[(size_t)sc * nb + b1] + W.fric[(size_t)sc * nb + b2]);
        cop[(size_t)(o + 7) * MX + c] = 0.0;  // h does not enter the backward system (lcp.py:176-183)
    }
    if (lane == 0) A.bw_nc[sc] = v.nc_k;
}

__global__ void __launch_bounds__(64) bwd_post_kernel(DssWorld W_arg, DssAdjoint A_arg)
{
    static_assert(sizeof(DssWorld) % 8 == 0, "the adjoint descriptor follows the world descriptor without padding");
    DSS_KERNARG_REF(DssWorld, W, W_arg);
    DSS_KERNARG_REF_AT(DssAdjoint, A, A_arg, sizeof(DssWorld));
    const int sc = blockIdx.x, lane = threadIdx.x, nb = W.nb, MX = W.maxc;
    if (!A.bw_active[sc]) return;
    const int k = A.cur_slot[sc];
    SlotView v;
    view_slot(W, sc, k, v);
    const int ND = W.fric_dirs / 2, NF = 3 * (1 + ND) + 8, o = 3 * (1 + ND);
    double *a_pose = A.a_pose + (size_t)sc * nb * 7, *a_vel = A.a_vel + (size_t)sc * nb * 6;
    double *a_geom = A.a_geom + (size_t)sc * 10 * MX, *cs = A.cscr + (size_t)sc * DSS_CSCR_ROWS * MX;
    const double *dcop = A.dcop + (size_t)sc * NF * MX, *dM = A.dMblk + (size_t)sc * nb * 36, *du = A.dpvec + (size_t)sc * 6 * nb;

    // contacts of sub-step k: adjoint of (dirs, p1, p2, mu, h_n) -> geometry, friction, restitution, velocities
    for (int c = lane; c < MX; c += 64) {
        if (c >= v.nc_k) { for (int i = 0; i < 10; ++i) a_geom[(size_t)i * MX + c] = 0.0; continue; }
        const int b1 = v.body_k[c], b2 = v.body_k[MX + c];
        const size_t i1 = (size_t)sc * nb + b1, i2 = (size_t)sc * nb + b2;
        double n[3], p1[3], p2[3], nbar[3], p1bar[3], p2bar[3];
        for (int i = 0; i < 3; ++i) {
            n[i] = v.geom_k[(size_t)i * MX + c]; p1[i] = v.geom_k[(size_t)(3 + i) * MX + c]; p2[i] = v.geom_k[(size_t)(6 + i) * MX + c];
            nbar[i] = dcop[(size_t)i * MX + c]; p1bar[i] = dcop[(size_t)(o + i) * MX + c]; p2bar[i] = dcop[(size_t)(o + 3 + i) * MX + c];
        }
        {   // friction directions D_q(n)
            typedef Dual<3> D;
            D nd[3], Dq[4][3];
            for (int i = 0; i < 3; ++i) { nd[i] = D(n[i]); nd[i].d[i] = 1.0; }
            friction_dirs(nd, ND, Dq);
            for (int s = 0; s < 3; ++s) {
                double acc = 0.0;
                for (int q = 0; q < ND; ++q) for (int i = 0; i < 3; ++i) acc += dcop[(size_t)(3 * (q + 1) + i) * MX + c] * Dq[q][i].d[s];
                nbar[s] += acc;
            }
        }
        const double mubar = dcop[(size_t)(o + 6) * MX + c], hbar = dcop[(size_t)(o + 7) * MX + c];
        const double *v1 = v.vel_k + 6 * b1, *v2 = v.vel_k + 6 * b2;
        double c1[3], c2[3], jv = 0.0;
        cross(p1, n, c1);
        cross(p2, n, c2);
        for (int i = 0; i < 3; ++i) jv += c1[i] * v1[i] + n[i] * v1[3 + i] - c2[i] * v2[i] - n[i] * v2[3 + i];
        const double rc = 0.5 * (W.restitution[i1] + W.restitution[i2]);
        const double jvbar = hbar * rc, rcbar = hbar * jv;
        // jv = n . ((w1 x p1) + u1 - (w2 x p2) - u2)
        double w1p[3], w2p[3], nw1[3], nw2[3];
        cross(v1, p1, w1p);
        cross(v2, p2, w2p);
        cross(n, v1, nw1);
        cross(n, v2, nw2);
        // (stop_contact_grad: h = Jc v is formed from a detached Jc -- no gradient to the geometry, the one to v stays)
        const double jg = (W.grad_flags & DSS_GRAD_STOP_CONTACT) ? 0.0 : jvbar;
        for (int i = 0; i < 3; ++i) {
            nbar[i] += jg * (w1p[i] + v1[3 + i] - w2p[i] - v2[3 + i]);
            p1bar[i] += jg * nw1[i];
            p2bar[i] -= jg * nw2[i];
            a_geom[(size_t)i * MX + c] = nbar[i];
        }
        for (int i = 0; i < 3; ++i) { a_geom[(size_t)(3 + i) * MX + c] = p1bar[i]; a_geom[(size_t)(6 + i) * MX + c] = p2bar[i]; }
        a_geom[(size_t)9 * MX + c] = 0.0;
        // per-contact pieces for the ordered per-body sums: velocity adjoints (12), mu, restitution
        for (int i = 0; i < 3; ++i) {
            cs[(size_t)i * MX + c] = jvbar * c1[i]; cs[(size_t)(3 + i) * MX + c] = jvbar * n[i];
            cs[(size_t)(6 + i) * MX + c] = -jvbar * c2[i]; cs[(size_t)(9 + i) * MX + c] = -jvbar * n[i];
        }
        cs[(size_t)12 * MX + c] = 0.5 * mubar;
        cs[(size_t)13 * MX + c] = 0.5 * rcbar;
    }
    __syncthreads();
    __shared__ double s_sums[64 * 8];
    {
        static constexpr int row0[8] = {0, 1, 2, 3, 4, 5, 12, 13}, row1[8] = {6, 7, 8, 9, 10, 11, 12, 13};
        contact_sums<8>(v.body_k, MX, v.nc_k, nb, cs, row0, row1, s_sums);
    }
    if (lane < nb) {
        const size_t bi = (size_t)sc * nb + lane;
        const double *vk = v.vel_k + 6 * lane, *ub = du + 6 * lane;
        double Iw[9], av[6], Iwbar[9];
        world_inertia(v.pose_k + 7 * lane, W.inertia + bi * 9, Iw);
        const double m = W.mass[bi];
        // u = M v + dt f
        for (int r = 0; r < 3; ++r) {
            av[r] = Iw[r] * ub[0] + Iw[3 + r] * ub[1] + Iw[6 + r] * ub[2];   // Iw^T ubar
            av[3 + r] = m * ub[3 + r];
        }
        double mbar = ub[3] * vk[3] + ub[4] * vk[4] + ub[5] * vk[5] + dM[36 * lane + 21] + dM[36 * lane + 28] + dM[36 * lane + 35];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) Iwbar[3 * r + c] = ub[r] * vk[c] + dM[36 * lane + 6 * r + c];
        for (int i = 0; i < 6; ++i) A.g_fext[bi * 6 + i] += v.dt * ub[i];
        for (int i = 0; i < 6; ++i) av[i] += s_sums[8 * lane + i];
        const double fricb = s_sums[8 * lane + 6], restb = s_sums[8 * lane + 7];
        for (int i = 0; i < 6; ++i) a_vel[6 * lane + i] = av[i];
        A.g_mass[bi] += mbar;
        A.g_fric[bi] += fricb;
        A.g_rest[bi] += restb;
        // Iw = R Ib R^T : Ib_bar = R^T Iw_bar R (linear), q_bar by duals
        double R[9], t[9];
        quat_to_mat(v.pose_k + 7 * lane, R);
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) t[3 * a + b] = R[a] * Iwbar[b] + R[3 + a] * Iwbar[3 + b] + R[6 + a] * Iwbar[6 + b];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) A.g_inertia[bi * 9 + 3 * a + b] += t[3 * a] * R[b] + t[3 * a + 1] * R[3 + b] + t[3 * a + 2] * R[6 + b];
        {
            typedef Dual<4> D;
            D q[4], Ib[9], out[9];
            for (int i = 0; i < 4; ++i) { q[i] = D(v.pose_k[7 * lane + i]); q[i].d[i] = 1.0; }
            for (int i = 0; i < 9; ++i) Ib[i] = D(W.inertia[bi * 9 + i]);
            world_inertia(q, Ib, out);
            for (int s = 0; s < 4; ++s) { double acc = 0.0; for (int e = 0; e < 9; ++e) acc += Iwbar[e] * out[e].d[s]; a_pose[7 * lane + s] += acc; }
        }
    }
    {   // u = M v + dt f : d/d(dt) = ubar . f ;  a sub-step whose dt_ was formed with last_dt hands -dt_bar back
        double part = 0.0;
        if (lane < nb) for (int i = 0; i < 6; ++i) part += du[6 * lane + i] * W.fext[((size_t)sc * nb + lane) * 6 + i];
        part = wave_sum(part);
        if (lane == 0) {
            const double dtbar = A.a_dt[sc] + part;
            A.a_last_dt[sc] = (W.tp_flags[(size_t)k * W.B + sc] & 2) ? -dtbar : 0.0;
            A.cur_slot[sc] = k - 1;
        }
    }
}

}  // namespace

#if DSS_ALL_SHAPES
namespace dss {
int launch_igr_list(const DssIgrNet &N, const double *pts, const int *lat_idx, const double *latents, int lat_stride,
                    const int *n_dev, int n_cap, int mode, double *sdf, double *grad, hipStream_t stream, int est);
void launch_bwd_pre_all(const DssWorld &W, const DssAdjoint &A, hipStream_t stream)
{
    if (W.igr.W0 && A.igr_bw_idx) {
        const int cap = W.B * 2 * W.maxc;
        (void)hipMemsetAsync(A.igr_bw_n, 0, sizeof(int), stream);
        hipLaunchKernelGGL(bwd_igr_prep_kernel, dim3(W.B), dim3(64), 0, stream, W, A);
        launch_igr_list(W.igr, A.igr_bw_pts, A.igr_bw_lat, W.shape_prm, 3, A.igr_bw_n, cap, DSS_IGR_XYZ, A.igr_bw_sdf, A.igr_bw_grad, stream, W.B * 8);
        launch_igr_list(W.igr, A.igr_bw_pts, A.igr_bw_lat, W.shape_prm, 3, A.igr_bw_n, cap, DSS_IGR_LATENT, A.igr_bw_sdf + cap,
                        A.igr_bw_grad + (size_t)cap * 3, stream, W.B * 8);
    }
    hipLaunchKernelGGL(bwd_pre_kernel, dim3(W.B), dim3(64), 0, stream, W, A);
}
}  // namespace dss
#else
namespace dss {
void launch_bwd_pre_all(const DssWorld &W, const DssAdjoint &A, hipStream_t stream);
int lcp_contact_backward_rows(const double *Mblk, const double *A, const double *cop, const int *cbody, const int *nc,
                              const int *active, int B, int nb, int neq, int maxc, int fric_dirs, const double *x,
                              const double *lam, const double *slack, const double *nu, const double *dl_dx, double *dMblk,
                              double *dpvec, double *dcop, double *dA, double *db, int rows, const int *slot, void *stream);     // lcp_contact.hip
}

extern "C" {

size_t dss_adjoint_sizeof(void) { return sizeof(DssAdjoint); }

int dss_step_backward(const DssWorld *W, const DssAdjoint *A, void *stream_)
{
    if (!W || !A || !W->tp_pose) return DSS_E_BADARG;
    hipStream_t stream = (hipStream_t)stream_;
    if (W->shape_rare) dss::launch_bwd_pre_all(*W, *A, stream);
    else hipLaunchKernelGGL(bwd_pre_kernel, dim3(W->B), dim3(64), 0, stream, *W, *A);
    // rows of G whose dependence on the contact geometry carries gradient: 1 normal (Jc), 2 friction (Jf)
    const int rows = ((W->grad_flags & DSS_GRAD_STOP_CONTACT) ? 0 : 1) | ((W->grad_flags & DSS_GRAD_STOP_FRICTION) ? 0 : 2);
    int rc = dss::lcp_contact_backward_rows(W->Mblk, W->Je, W->cop, W->cop_body, A->bw_nc, A->bw_active, W->B, W->nb, W->neq,
                                            W->maxc, W->fric_dirs, W->x, W->tp_lam, W->tp_slack, W->nu, A->a_x, A->dMblk, A->dpvec,
                                            A->dcop, nullptr, nullptr, rows, A->cur_slot, stream_);      // (multipliers read from the tape in place)
    if (rc) return rc;
    hipLaunchKernelGGL(bwd_post_kernel, dim3(W->B), dim3(64), 0, stream, *W, *A);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}

}  // extern "C"
#endif
