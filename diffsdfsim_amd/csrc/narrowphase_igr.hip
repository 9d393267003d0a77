// narrowphase_igr.hip -- contact detection for pairs with a NEURAL SDF body (gfx950).
//
// Restates FWContactHandler (sdf_physics/physics3d/contacts.py:39-272) for SDF3D bodies whose sdf_func is an IGR network
// (decode_igr, utils.py:330-350; query_sdfs with the autograd input gradient, bodies.py:721-760).  A network evaluation is
// 115 k multiply-adds per point: it belongs on the matrix cores, in batches, not in the lane-per-candidate code of
// narrowphase.hip.  So a work item (scene, directed pair a -> b) is a small state machine that runs from one batch of SDF
// queries to the next ("round"):
//
//   round r:   igr_query_kernel (igr_mlp.hip) evaluates the two point lists the items filled in round r-1 -- values only
//              (candidate test, Laplacian probes) and values + d/dxyz -- for ALL items of ALL scenes at once: full MFMA tiles
//              however few points one item contributes (a Frank-Wolfe iteration moves 2-6 points per item);
//              igr_advance_kernel: every unfinished item (one 256-thread workgroup each) consumes its results, advances as
//              far as it can without another network value, and appends its next queries to the lists of round r+1.
//
//   SCAN   candidate faces of a's mesh (culling boxes, centroid pre-test); b neural: phi_b at the centroids   contacts.py:44-52
//   CAND   phi_b < rad + eps -> candidates; queries at their three vertices (+ gradient at the centroid)      contacts.py:52-61
//   INIT   start vertex = arg min phi_b; |grad| > 1e-12 test; first Frank-Wolfe evaluation                     contacts.py:57-61
//   FW     <= 32 iterations, one round each (float32 step sizes, early exits as in the reference)              contacts.py:63-82
//   PROJ   pull the point onto a's surface (phi_a, grad_a at the barycentric point), keep phi_b <= eps        contacts.py:84-94
//   G1-G4  contact geometry: Newton step on a, phi/grad of both bodies, the twelve Laplacian probes           contacts.py:161-214
//   then the thinning and output stages shared with narrowphase.hip (np_common.h)                           contacts.py:97-158
//
// A pair whose bodies are both analytic never comes here; a stage whose query body is analytic evaluates it in place and
// falls through to the next stage.  Query slots are reserved with one atomic per item and list; results depend on the
// point alone, so the order in which items reserve is immaterial and the contact sets are deterministic.
#define DSS_ALL_SHAPES 1
#include "np_common.h"

namespace dss {
int launch_igr_pair(const DssIgrNet &N, const double *pts_v, const int *lat_v, const int *n_v, double *sdf_v, const double *pts_g,
                    const int *lat_g, const int *n_g, double *sdf_g, double *grad_g, const double *latents, int lat_stride, int n_cap,
                    hipStream_t stream, int est_v, int est_g);
}

namespace {
using G = BlockGroup;

enum { ST_SCAN = 0, ST_CAND, ST_INIT, ST_FW, ST_PROJ, ST_PROJ2, ST_PROJ3, ST_G2, ST_G3, ST_G4, ST_DONE };
enum { H_STATE = 0, H_NCAND, H_NCON, H_ITER, H_OVER, H_QV, H_QG };
enum { L_VALUE = 0, L_GRAD = 1 };

#define CB(f, k) cb[(size_t)(f) * MC + (k)]

__device__ inline bool in_cube(const double *p, double s) { return fabs(p[0]) <= s && fabs(p[1]) <= s && fabs(p[2]) <= s; }

// the two query lists of one round
struct Lists {
    double *pts[2];
    int *lat[2];
    int *tag[2];
    const double *sdf[2];
    const double *grad;
    int *count;   // [2]
    int cap;
};
__device__ inline Lists lists_of(const DssWorld &W, int round)
{
    const int set = round & 1;
    Lists L;
    for (int l = 0; l < 2; ++l) {
        L.pts[l] = W.igr_qpts + (size_t)(2 * set + l) * W.igr_qcap * 3;
        L.lat[l] = W.igr_qlat + (size_t)(2 * set + l) * W.igr_qcap;
        L.tag[l] = W.igr_qtag + (size_t)(2 * set + l) * W.igr_qcap;
        L.sdf[l] = W.igr_qsdf + (size_t)(2 * set + l) * W.igr_qcap;
    }
    L.grad = W.igr_qgrad + (size_t)set * W.igr_qcap * 3;
    L.count = W.igr_qn + 2 * round;
    L.cap = W.igr_qcap;
    return L;
}
// reserve `n` slots of list `l` for the item (one atomic); -1 when the list is full
__device__ inline int reserve(const Lists &L, int l, int n, ScratchT<G> &S)
{
    if (G::tid() == 0) {
        int b = n > 0 ? atomicAdd(L.count + l, n) : 0;
        if (b + n > L.cap) { atomicAdd(L.count + l, -n); b = -1; }   // (the round's length stays within the lists)
        S.wave_tot[0] = b;
    }
    G::sync();
    const int base = S.wave_tot[0];
    G::sync();
    return base;
}
// SDF3D.query_sdfs (bodies.py:721-751): a point outside the query cube is not evaluated (phi = scale, grad = 0); inside, the
// network sees pts / scale.  A slot is written in either case (a dummy for outside points) so that slots stay computable.
__device__ inline void put(const Lists &L, int l, int idx, const double *pt, double scale, int lat)
{
    double u[3] = {0.0, 0.0, 0.0};
    if (in_cube(pt, scale)) div3(pt, scale, u);
    for (int i = 0; i < 3; ++i) L.pts[l][(size_t)idx * 3 + i] = u[i];
    L.lat[l][idx] = lat;
}
__device__ inline double take_value(const Lists &L, int l, int idx, const double *pt, double scale)
{
    return in_cube(pt, scale) ? L.sdf[l][idx] * scale : scale;
}
__device__ inline void take_grad(const Lists &L, int idx, const double *pt, double scale, double &phi, double *g)
{
    if (!in_cube(pt, scale)) { phi = scale; g[0] = g[1] = g[2] = 0.0; return; }
    phi = L.sdf[L_GRAD][idx] * scale;
    const double raw[3] = {L.grad[(size_t)idx * 3], L.grad[(size_t)idx * 3 + 1], L.grad[(size_t)idx * 3 + 2]};
    normalize(raw, g);      // F.normalize(grads_ov) (bodies.py:741)
}

struct Cand { double pqr[9], x[3], abc[3]; };
__device__ inline double vtx(const Cand &c, int bi, int i) { return bi == 0 ? c.pqr[i] : (bi == 1 ? c.pqr[3 + i] : c.pqr[6 + i]); }
// one Frank-Wolfe evaluation (contacts.py:64-73) from phi, grad at the current point.  gamma is python_float * bool_tensor,
// which torch promotes to float32: replicated (see narrowphase.hip)
__device__ inline void fw_eval(const Cand &c, double phi, const double *g, int iter, double tol, float &gm, int &bi, int &pen)
{
    double bestd = INFINITY; bi = 0;
    for (int v = 0; v < 3; ++v) {
        const double d = c.pqr[3 * v] * g[0] + c.pqr[3 * v + 1] * g[1] + c.pqr[3 * v + 2] * g[2];
        if (d < bestd) { bestd = d; bi = v; }
    }
    const double impr = (c.x[0] - vtx(c, bi, 0)) * g[0] + (c.x[1] - vtx(c, bi, 1)) * g[1] + (c.x[2] - vtx(c, bi, 2)) * g[2];
    gm = (fabs(impr) > tol) ? (float)(2.0 / (iter + 2.0)) : 0.0f;
    pen = phi < -tol;
}
__device__ inline void fw_apply(Cand &c, float g32, int bi)
{
    const double gm = (double)g32, om = (double)(1.0f - g32);
    for (int i = 0; i < 3; ++i) { c.x[i] = om * c.x[i] + gm * vtx(c, bi, i); c.abc[i] *= om; }
    for (int i = 0; i < 3; ++i) if (i == bi) c.abc[i] += gm;
}

__device__ void advance(const DssWorld &W, ScratchT<G> &S, int it, int round, int last_round)
{
    int *hdr = W.igr_hdr + (size_t)it * DSS_IGR_HDR;
    const int tid = G::tid();
    int state = round == 0 ? ST_SCAN : hdr[H_STATE];
    if (state == ST_DONE) return;
    const int item = W.igr_list[it];
    const int np = npairs_of(W.nb), sc = item / np, dp = item % np;
    int a, b;
    pair_of(dp, W.nb, a, b);
    int *pc_count = W.pc_count + (size_t)sc * np + dp;
    BodyD A, Bd;
    load_body(W, sc, a, A);
    load_body(W, sc, b, Bd);
    const bool Aigr = A.g.shape.type == SHAPE_IGR, Bigr = Bd.g.shape.type == SHAPE_IGR;
    const double sA = A.g.shape.scale, sB = Bd.g.shape.scale;
    const int latA = sc * W.nb + a, latB = sc * W.nb + b, nq = (Aigr ? 1 : 0) + (Bigr ? 1 : 0);
    const int MC = W.max_cand;
    int *__restrict__ cface = W.igr_cface + (size_t)it * 3 * MC, *__restrict__ kface = cface + MC, *__restrict__ qrank = cface + 2 * MC;
    int *__restrict__ cstate = W.igr_cstate + (size_t)it * MC;
    double *__restrict__ cb = W.igr_cbuf + (size_t)it * DSS_CAND_FIELDS * MC;
    const double *__restrict__ m_verts = W.verts, *__restrict__ m_fcent = W.fcent;
    const int *__restrict__ m_faces = W.faces;
    const Lists R = lists_of(W, round), Q = lists_of(W, round + 1);     // results of this round, queries of the next
    int ncand = round == 0 ? 0 : hdr[H_NCAND], ncon = round == 0 ? 0 : hdr[H_NCON], iter = round == 0 ? 0 : hdr[H_ITER];
    int over = round == 0 ? 0 : hdr[H_OVER], qv = hdr[H_QV], qg = hdr[H_QG];
    G::sync();   // everybody has read the header before anybody rewrites it

    auto load_c = [&](Cand &c, int k) {
        for (int i = 0; i < 9; ++i) c.pqr[i] = CB(i, k);
        for (int i = 0; i < 3; ++i) { c.x[i] = CB(9 + i, k); c.abc[i] = CB(12 + i, k); }
    };
    auto store_c = [&](const Cand &c, int k) {
        for (int i = 0; i < 3; ++i) { CB(9 + i, k) = c.x[i]; CB(12 + i, k) = c.abc[i]; }
    };
    // barycentric point of candidate / contact k on its triangle, in a's body frame
    auto bary_point = [&](int face, const double *abc, double *o) {
        const int *fv = m_faces + (size_t)(A.foff + face) * 3;
        o[0] = o[1] = o[2] = 0.0;
        for (int v = 0; v < 3; ++v) {
            const double *vp = m_verts + (size_t)(A.voff + fv[v]) * 3;
            for (int i = 0; i < 3; ++i) o[i] += vp[i] * abc[v];
        }
    };
    auto finish = [&](int count) {      // the item ends with `count` contacts already written (0: none)
        if (tid == 0) { if (count >= 0) *pc_count = count; if (over) atomicOr(W.overflow + sc, over); }
        state = ST_DONE;
    };
    bool wait = false;     // the next stage needs the results of a query round

    while (state != ST_DONE && !wait) {
        switch (state) {
        case ST_SCAN: {
            // ---- candidate faces (contacts.py:44-52) in ascending face order ------------------------------------------
            double Ra[9], Rb[9], R12[9], t12[3];
            quat_to_mat(A.g.q, Ra);
            quat_to_mat(Bd.g.q, Rb);
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) R12[3 * i + j] = Rb[i] * Ra[j] + Rb[3 + i] * Ra[3 + j] + Rb[6 + i] * Ra[6 + j];
                t12[i] = Rb[i] * (A.g.pos[0] - Bd.g.pos[0]) + Rb[3 + i] * (A.g.pos[1] - Bd.g.pos[1]) + Rb[6 + i] * (A.g.pos[2] - Bd.g.pos[2]);
            }
            Region reg;
            region_of(A.g, Bd.g, 1e-9, reg);
            const double *fbox = W.fch_box + (size_t)W.mesh_fch_off[A.mesh] * 6;
            constexpr int RUN = 256;
            static_assert(RUN == G::BT, "one culling box per round of the group");
            // the centroid, bounding radius and the reference's own candidate test of one face (contacts.py:42-52)
            auto face_geom = [&](int f, double pqr[3][3], double *x, double &rad) {
                const int *fv = m_faces + (size_t)(A.foff + f) * 3;
                x[0] = x[1] = x[2] = 0.0; rad = 0.0;
                for (int k = 0; k < 3; ++k) {
                    to_frame(A.g, Bd.g, m_verts + (size_t)(A.voff + fv[k]) * 3, pqr[k]);
                    for (int i = 0; i < 3; ++i) x[i] += pqr[k][i];
                }
                div3(x, 3.0, x);
                for (int k = 0; k < 3; ++k) {
                    const double d[3] = {x[0] - pqr[k][0], x[1] - pqr[k][1], x[2] - pqr[k][2]};
                    const double r = t_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                    if (r > rad) rad = r;
                }
            };
            // b neural: every face whose centroid lies in b's query cube needs phi_b there before it can be judged.  Such
            // "tentative" faces can be many (all of a floor's faces under the body), so they live in the query list itself
            // (point + face id as the tag), not in the item's candidate scratch; two passes: count, then reserve and write.
            // b analytic: the runs of 256 faces that can hold a candidate are listed first, one culling box per thread.  Beyond
            // the query-cube test a run is dropped if b's surface is out of reach of every face in it: b's SDF is an exact
            // distance (box / sphere / cylinder / ...: 1-Lipschitz), the run's box holds every face's bounding sphere, so
            // phi_b(centroid) >= phi_b(box centre) - |half diagonal| and rad <= the smallest half extent; a face with
            // phi_b >= rad + eps is no candidate (contacts.py:52).  A neural body's whole mesh lies in the query cube of a floor
            // it is nowhere near -- 60 k faces read and tested per item and attempt, for nothing, without this.
            const int nch = (A.nf + RUN - 1) / RUN;
            int nrun = -1;
            if (!Bigr && nch <= G::HCAP) {
                nrun = 0;
                for (int cb0 = 0; cb0 < nch; cb0 += G::BT) {
                    const int ch = cb0 + tid;
                    int keep = 0;
                    if (ch < nch) {
                        const double *bx = fbox + (size_t)ch * 6;
                        keep = box_hits(reg, bx);
                        if (keep && Bd.g.shape.type != SHAPE_GRID && Bd.g.shape.type != SHAPE_BOWL) {
                            double m[3], e2 = 0.0, emin = INFINITY;
                            for (int i = 0; i < 3; ++i) {
                                m[i] = 0.5 * (bx[i] + bx[3 + i]);
                                const double e = 0.5 * (bx[3 + i] - bx[i]);
                                e2 += e * e; emin = fmin(emin, e);
                            }
                            double pu[3], u, gdum[3];
                            for (int i = 0; i < 3; ++i) pu[i] = (R12[3 * i] * m[0] + R12[3 * i + 1] * m[1] + R12[3 * i + 2] * m[2] + t12[i]) / sB;
                            sdf_unit(Bd.g.shape, pu, u, gdum, false);
                            if (u * sB - sqrt(e2) >= emin + W.eps + 1e-9 * (1.0 + sB)) keep = 0;
                        }
                    }
                    const int slot = compact_slot(keep, nrun, S);
                    if (slot >= 0) S.hidx[slot] = ch;
                }
                G::sync();
            }
            int ntent = 0;
            for (int pass = 0; pass < (Bigr ? 2 : 1); ++pass) {
                int cnt = 0;
                for (int it = 0; it < (nrun >= 0 ? nrun : nch); ++it) {
                    const int base = (nrun >= 0 ? S.hidx[it] : it) * RUN;
                    if (nrun < 0 && !box_hits(reg, fbox + (size_t)(base / RUN) * 6)) continue;     // (uniform over the group)
                    const int f = base + tid;
                    int flag = 0;
                    double pqr[3][3], x[3] = {0, 0, 0}, rad = 0.0;
                    if (f < A.nf) {
                        // cheap pre-test: the pose-invariant centroid lies in b's query cube (+ margin)
                        const double *c = m_fcent + (size_t)(A.foff + f) * 3;
                        double c2[3];
                        for (int i = 0; i < 3; ++i) c2[i] = R12[3 * i] * c[0] + R12[3 * i + 1] * c[1] + R12[3 * i + 2] * c[2] + t12[i];
                        const double lim = sB + 1e-9 * (1.0 + sB);
                        if (fabs(c2[0]) <= lim && fabs(c2[1]) <= lim && fabs(c2[2]) <= lim) {
                            face_geom(f, pqr, x, rad);
                            if (Bigr) flag = in_cube(x, sB);        // outside: phi = scale, grad = 0 -> never a candidate
                            else {
                                double phi, g[3];
                                const bool inc = query_sdf(Bd.g.shape, x, phi, g, true);
                                const double gn = t_sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
                                flag = inc && (phi < rad + W.eps) && (gn > 1e-12);
                            }
                        }
                    }
                    if (!G::any(flag)) continue;
                    const int slot = compact_slot(flag, cnt, S);
                    if (Bigr) {
                        if (pass == 1 && slot >= 0) { put(Q, L_VALUE, qv + slot, x, sB, latB); Q.tag[L_VALUE][qv + slot] = f; }
                    } else if (slot >= 0 && slot < MC) {
                        cface[slot] = f;
                        for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) CB(3 * k + i, slot) = pqr[k][i];
                    }
                }
                G::sync();
                if (Bigr && pass == 0) {
                    ntent = cnt;
                    if (ntent == 0) break;
                    qv = reserve(Q, L_VALUE, ntent, S);
                    if (qv < 0) { over |= 32; ntent = 0; break; }
                } else if (!Bigr) {
                    ncand = cnt;
                    if (ncand > MC) { over |= 1; ncand = MC; }
                }
            }
            if (Bigr) {
                if (ntent == 0) { finish(0); break; }
                ncand = ntent;        // (the header's candidate count carries the number of tentative faces to the next stage)
                state = ST_CAND; wait = true;
            } else {
                // b analytic: the whole Frank-Wolfe search needs no network value
                for (int k = tid; k < ncand; k += G::BT) {
                    Cand c; load_c(c, k);
                    double best = INFINITY; int bi = 0;
                    for (int v = 0; v < 3; ++v) {
                        double phi, g[3];
                        query_sdf(Bd.g.shape, c.pqr + 3 * v, phi, g, false);
                        if (phi < best) { best = phi; bi = v; }
                    }
                    for (int i = 0; i < 3; ++i) { c.x[i] = vtx(c, bi, i); c.abc[i] = (i == bi) ? 1.0 : 0.0; }
                    store_c(c, k);
                    cstate[k] = 0;
                }
                // Iteration 0 looks at every candidate; a candidate whose |improvement| <= tol is frozen for good (gamma = 0: x no
                // longer changes, every later evaluation repeats this one), so afterwards only the movers -- typically a handful
                // on the rim of a contact patch, out of hundreds of candidates -- are walked, through a packed list.
                int nlist = ncand;          // entries of the list the iterations walk: all candidates, then the movers
                bool packed = false;
                for (int itn = 0; itn < 32; ++itn) {
                    int moving = 0, anyp = 0;
                    for (int j = tid; j < nlist; j += G::BT) {
                        const int k = packed ? S.hidx[j] : j;
                        if (cstate[k] != 0) continue;
                        Cand c; load_c(c, k);
                        double phi, g[3];
                        query_sdf(Bd.g.shape, c.x, phi, g, true);
                        float gm; int bi, pen;
                        fw_eval(c, phi, g, itn, W.tol, gm, bi, pen);
                        CB(25, k) = (double)gm; CB(26, k) = (double)bi;
                        moving |= gm != 0.0f; anyp |= pen;
                    }
                    const int mv = G::any(moving), pn = G::any(anyp);
                    if (!mv || pn) break;           // all gamma == 0, or a penetrating point (contacts.py:74-77)
                    for (int j = tid; j < nlist; j += G::BT) {
                        const int k = packed ? S.hidx[j] : j;
                        if (cstate[k] != 0) continue;
                        const float gm = (float)CB(25, k);
                        if (gm == 0.0f) { cstate[k] = -1; continue; }      // froze: x no longer changes
                        Cand c; load_c(c, k); fw_apply(c, gm, (int)CB(26, k)); store_c(c, k);
                    }
                    if (itn == 0) {
                        G::sync();
                        int nm = 0;
                        for (int base = 0; base < ncand; base += G::BT) {
                            const int k = base + tid;
                            const int slot = compact_slot(k < ncand && cstate[k] == 0, nm, S);
                            if (slot >= 0 && slot < G::HCAP) S.hidx[slot] = k;
                        }
                        G::sync();
                        if (nm <= G::HCAP) { nlist = nm; packed = true; }      // (more movers than the list holds: keep walking them all)
                    }
                }
                G::sync();
                state = ST_PROJ;
            }
            break;
        }
        case ST_CAND: {
            // ---- phi_b(centroid) < rad + eps (contacts.py:52); survivors keep their ascending order ---------------------
            const int nt = ncand;
            ncand = 0;
            for (int base = 0; base < nt; base += G::BT) {
                const int k = base + tid;
                int flag = 0, f = 0;
                double pqr[3][3], x[3] = {0, 0, 0}, rad = 0.0;
                if (k < nt) {
                    f = R.tag[L_VALUE][qv + k];
                    const int *fv = m_faces + (size_t)(A.foff + f) * 3;
                    for (int v = 0; v < 3; ++v) {       // the same arithmetic as in the scan: the same centroid, bit for bit
                        to_frame(A.g, Bd.g, m_verts + (size_t)(A.voff + fv[v]) * 3, pqr[v]);
                        for (int i = 0; i < 3; ++i) x[i] += pqr[v][i];
                    }
                    div3(x, 3.0, x);
                    for (int v = 0; v < 3; ++v) {
                        const double d[3] = {x[0] - pqr[v][0], x[1] - pqr[v][1], x[2] - pqr[v][2]};
                        const double r = t_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                        if (r > rad) rad = r;
                    }
                    const double phi = take_value(R, L_VALUE, qv + k, x, sB);
                    flag = phi < rad + W.eps;
                }
                if (!G::any(flag)) continue;
                const int slot = compact_slot(flag, ncand, S);
                if (slot >= 0 && slot < MC) {
                    cface[slot] = f;
                    for (int v = 0; v < 3; ++v) for (int i = 0; i < 3; ++i) CB(3 * v + i, slot) = pqr[v][i];
                    for (int i = 0; i < 3; ++i) CB(9 + i, slot) = x[i];
                }
            }
            if (ncand > MC) { over |= 1; ncand = MC; }
            G::sync();
            if (ncand == 0) { finish(0); break; }
            qv = reserve(Q, L_VALUE, 3 * ncand, S);
            qg = reserve(Q, L_GRAD, ncand, S);
            if (qv < 0 || qg < 0) { over |= 32; finish(0); break; }
            for (int k = tid; k < ncand; k += G::BT) {
                double pt[3];
                for (int i = 0; i < 3; ++i) pt[i] = CB(9 + i, k);
                put(Q, L_GRAD, qg + k, pt, sB, latB);                 // gradient at the centroid: the |grad| > 1e-12 test
                for (int v = 0; v < 3; ++v) {
                    for (int i = 0; i < 3; ++i) pt[i] = CB(3 * v + i, k);
                    put(Q, L_VALUE, qv + 3 * k + v, pt, sB, latB);
                }
            }
            state = ST_INIT; wait = true;
            break;
        }
        case ST_INIT: {
            // ---- start vertex (contacts.py:57-61) -----------------------------------------------------------------------
            for (int k = tid; k < ncand; k += G::BT) {
                Cand c; load_c(c, k);
                double phc, gc[3];
                take_grad(R, qg + k, c.x, sB, phc, gc);
                const double gn = t_sqrt(gc[0] * gc[0] + gc[1] * gc[1] + gc[2] * gc[2]);
                double best = INFINITY; int bi = 0;
                for (int v = 0; v < 3; ++v) {
                    const double phi = take_value(R, L_VALUE, qv + 3 * k + v, c.pqr + 3 * v, sB);
                    if (phi < best) { best = phi; bi = v; }
                }
                for (int i = 0; i < 3; ++i) { c.x[i] = vtx(c, bi, i); c.abc[i] = (i == bi) ? 1.0 : 0.0; }
                store_c(c, k);
                cstate[k] = gn > 1e-12 ? 0 : -3;      // -3: not a candidate after all (cand_mask, contacts.py:52)
                qrank[k] = k;
            }
            G::sync();
            qg = reserve(Q, L_GRAD, ncand, S);
            if (qg < 0) { over |= 32; finish(0); break; }
            for (int k = tid; k < ncand; k += G::BT) {
                const double x[3] = {CB(9, k), CB(10, k), CB(11, k)};
                put(Q, L_GRAD, qg + k, x, sB, latB);
            }
            iter = 0;
            state = ST_FW; wait = true;
            break;
        }
        case ST_FW: {
            // ---- one Frank-Wolfe iteration: the evaluation came back, vote, move, ask for the next (contacts.py:63-82) ---
            int moving = 0, anyp = 0;
            for (int k = tid; k < ncand; k += G::BT) {
                if (cstate[k] != 0) continue;
                Cand c; load_c(c, k);
                double phi, g[3];
                take_grad(R, qg + qrank[k], c.x, sB, phi, g);
                float gm; int bi, pen;
                fw_eval(c, phi, g, iter, W.tol, gm, bi, pen);
                CB(25, k) = (double)gm; CB(26, k) = (double)bi;
                moving |= gm != 0.0f; anyp |= pen;
            }
            const int mv = G::any(moving), pn = G::any(anyp);
            if (!mv || pn) { state = ST_PROJ; break; }
            for (int k = tid; k < ncand; k += G::BT) {
                if (cstate[k] != 0) continue;
                const float gm = (float)CB(25, k);
                if (gm == 0.0f) { cstate[k] = -1; continue; }
                Cand c; load_c(c, k); fw_apply(c, gm, (int)CB(26, k)); store_c(c, k);
            }
            G::sync();
            if (++iter >= 32) { state = ST_PROJ; break; }
            int nmov = 0;
            for (int base = 0; base < ncand; base += G::BT) {
                const int k = base + tid;
                const int mvk = k < ncand && cstate[k] == 0;
                const int slot = compact_slot(mvk, nmov, S);
                if (slot >= 0) qrank[k] = slot;
            }
            G::sync();
            qg = reserve(Q, L_GRAD, nmov, S);
            if (qg < 0) { over |= 32; finish(0); break; }
            for (int k = tid; k < ncand; k += G::BT) {
                if (cstate[k] != 0) continue;
                const double x[3] = {CB(9, k), CB(10, k), CB(11, k)};
                put(Q, L_GRAD, qg + qrank[k], x, sB, latB);
            }
            wait = true;
            break;
        }
        case ST_PROJ: {
            // ---- pull onto body a's surface (contacts.py:84-88): phi_a, grad_a at the barycentric point -----------------
            if (Aigr) {
                qg = reserve(Q, L_GRAD, ncand, S);
                if (qg < 0) { over |= 32; finish(0); break; }
                for (int k = tid; k < ncand; k += G::BT) {
                    const double abc[3] = {CB(12, k), CB(13, k), CB(14, k)};
                    double xb1[3];
                    bary_point(cface[k], abc, xb1);
                    put(Q, L_GRAD, qg + k, xb1, sA, latA);
                }
                wait = true;
            }
            state = ST_PROJ2;
            break;
        }
        case ST_PROJ2: {
            double qrel[4], qbi[4];
            quat_inv(Bd.g.q, qbi);
            quat_mul(qbi, A.g.q, qrel);
            for (int k = tid; k < ncand; k += G::BT) {
                const double abc[3] = {CB(12, k), CB(13, k), CB(14, k)};
                double xb1[3], phi1, g1[3], gr[3];
                bary_point(cface[k], abc, xb1);
                if (Aigr) take_grad(R, qg + k, xb1, sA, phi1, g1);
                else query_sdf(A.g.shape, xb1, phi1, g1, true);
                quat_apply(qrel, g1, gr);
                for (int i = 0; i < 3; ++i) CB(9 + i, k) = CB(9 + i, k) - phi1 * gr[i];
            }
            G::sync();
            if (Bigr) {
                qv = reserve(Q, L_VALUE, ncand, S);
                if (qv < 0) { over |= 32; finish(0); break; }
                for (int k = tid; k < ncand; k += G::BT) {
                    const double x[3] = {CB(9, k), CB(10, k), CB(11, k)};
                    put(Q, L_VALUE, qv + k, x, sB, latB);
                }
                wait = true;
            }
            state = ST_PROJ3;
            break;
        }
        case ST_PROJ3: {
            // ---- keep phi_b <= eps (contacts.py:89-94); then the first query of the contact geometry --------------------
            ncon = 0;
            for (int base = 0; base < ncand; base += G::BT) {
                const int k = base + tid;
                int flag = 0, f = 0;
                double abc[3] = {0, 0, 0};
                if (k < ncand && cstate[k] != -3) {
                    const double x[3] = {CB(9, k), CB(10, k), CB(11, k)};
                    double phi2, g2[3];
                    if (Bigr) phi2 = take_value(R, L_VALUE, qv + k, x, sB);
                    else query_sdf(Bd.g.shape, x, phi2, g2, false);
                    flag = phi2 <= W.eps;
                    f = cface[k];
                    for (int i = 0; i < 3; ++i) abc[i] = CB(12 + i, k);
                }
                if (!G::any(flag)) continue;
                const int slot = compact_slot(flag, ncon, S);
                if (slot >= 0) { kface[slot] = f; for (int i = 0; i < 3; ++i) CB(15 + i, slot) = abc[i]; }
            }
            G::sync();
            if (ncon == 0) { finish(0); break; }
            if (Aigr) {
                qg = reserve(Q, L_GRAD, ncon, S);
                if (qg < 0) { over |= 32; finish(0); break; }
                for (int j = tid; j < ncon; j += G::BT) {
                    const double abc[3] = {CB(15, j), CB(16, j), CB(17, j)};
                    double cp1[3];
                    bary_point(kface[j], abc, cp1);
                    put(Q, L_GRAD, qg + j, cp1, sA, latA);
                }
                wait = true;
            }
            state = ST_G2;
            break;
        }
        case ST_G2: {
            // ---- Newton step onto a's surface (contacts.py:165-171), the point in b's frame (:173-179) -----------------
            for (int j = tid; j < ncon; j += G::BT) {
                const double abc[3] = {CB(15, j), CB(16, j), CB(17, j)};
                double cp1[3], d1, n1[3], p1[3], rel[3], cp2[3];
                bary_point(kface[j], abc, cp1);
                if (Aigr) take_grad(R, qg + j, cp1, sA, d1, n1);
                else query_sdf(A.g.shape, cp1, d1, n1, true);
                for (int i = 0; i < 3; ++i) cp1[i] = cp1[i] - d1 * n1[i];
                quat_apply(A.g.q, cp1, p1);
                for (int i = 0; i < 3; ++i) rel[i] = (p1[i] + A.g.pos[i]) - Bd.g.pos[i];
                quat_apply_inv(Bd.g.q, rel, cp2);
                for (int i = 0; i < 3; ++i) { CB(3 + i, j) = cp1[i]; CB(6 + i, j) = cp2[i]; }
            }
            G::sync();
            qg = reserve(Q, L_GRAD, nq * ncon, S);
            if (qg < 0) { over |= 32; finish(0); break; }
            for (int j = tid; j < ncon; j += G::BT) {
                const double cp1[3] = {CB(3, j), CB(4, j), CB(5, j)}, cp2[3] = {CB(6, j), CB(7, j), CB(8, j)};
                if (Aigr) put(Q, L_GRAD, qg + nq * j, cp1, sA, latA);
                if (Bigr) put(Q, L_GRAD, qg + nq * j + (Aigr ? 1 : 0), cp2, sB, latB);
            }
            state = ST_G3; wait = true;
            break;
        }
        case ST_G3: {
            // ---- phi, normal of both bodies at the contact point; the Laplacian probes (contacts.py:171-196) ------------
            for (int j = tid; j < ncon; j += G::BT) {
                const double cp1[3] = {CB(3, j), CB(4, j), CB(5, j)}, cp2[3] = {CB(6, j), CB(7, j), CB(8, j)};
                double d1, n1[3], d2, n2[3];
                if (Aigr) take_grad(R, qg + nq * j, cp1, sA, d1, n1);
                else query_sdf(A.g.shape, cp1, d1, n1, true);
                if (Bigr) take_grad(R, qg + nq * j + (Aigr ? 1 : 0), cp2, sB, d2, n2);
                else query_sdf(Bd.g.shape, cp2, d2, n2, true);
                for (int i = 0; i < 3; ++i) { CB(9 + i, j) = n1[i]; CB(12 + i, j) = n2[i]; }
                CB(25, j) = d1; CB(26, j) = d2;
            }
            G::sync();
            qv = reserve(Q, L_VALUE, 6 * nq * ncon, S);
            if (qv < 0) { over |= 32; finish(0); break; }
            for (int j = tid; j < ncon; j += G::BT) {
                int o = qv + 6 * nq * j;
                for (int body = 0; body < 2; ++body) {
                    if (!(body == 0 ? Aigr : Bigr)) continue;
                    double pt[3] = {CB(3 + 3 * body, j), CB(4 + 3 * body, j), CB(5 + 3 * body, j)};
                    for (int i = 0; i < 3; ++i) {
                        const double save = pt[i];
                        pt[i] = save + 1e-3; put(Q, L_VALUE, o++, pt, body == 0 ? sA : sB, body == 0 ? latA : latB);
                        pt[i] = save - 1e-3; put(Q, L_VALUE, o++, pt, body == 0 ? sA : sB, body == 0 ? latA : latB);
                        pt[i] = save;
                    }
                }
            }
            state = ST_G4; wait = true;
            break;
        }
        case ST_G4: {
            // ---- which body's normal (contacts.py:184-202), contact points, penetration (:204-209) ----------------------
            int bad = 0;
            for (int j = tid; j < ncon; j += G::BT) {
                const double cp1[3] = {CB(3, j), CB(4, j), CB(5, j)}, cp2[3] = {CB(6, j), CB(7, j), CB(8, j)};
                const double n1[3] = {CB(9, j), CB(10, j), CB(11, j)}, n2[3] = {CB(12, j), CB(13, j), CB(14, j)};
                const double d1 = CB(25, j), d2 = CB(26, j);
                double lap[2];
                int o = qv + 6 * nq * j;
                for (int body = 0; body < 2; ++body) {
                    const double *cp = body == 0 ? cp1 : cp2;
                    const double phi0 = body == 0 ? d1 : d2, sc_ = body == 0 ? sA : sB;
                    if (!(body == 0 ? Aigr : Bigr)) { lap[body] = lap_probe(body == 0 ? A.g.shape : Bd.g.shape, cp, phi0, 1e-3); continue; }
                    double acc = 0.0, pt[3] = {cp[0], cp[1], cp[2]};
                    for (int i = 0; i < 3; ++i) {
                        const double save = pt[i];
                        pt[i] = save + 1e-3; const double pa = take_value(R, L_VALUE, o++, pt, sc_);
                        pt[i] = save - 1e-3; const double pb = take_value(R, L_VALUE, o++, pt, sc_);
                        pt[i] = save;
                        acc += pa - 2.0 * phi0 + pb;
                    }
                    lap[body] = acc;
                }
                const bool stable = fabs(lap[1]) < fabs(lap[0]);
                if (!stable) kface[j] |= DSS_FACE_NORMAL1;
                double n[3], t[3], p1[3], p2[3];
                if (stable) quat_apply(Bd.g.q, n2, n);
                else { quat_apply(A.g.q, n1, t); for (int i = 0; i < 3; ++i) n[i] = -t[i]; }
                for (int i = 0; i < 3; ++i) t[i] = cp2[i] - d2 * n2[i];
                quat_apply(Bd.g.q, t, p2);
                quat_apply(A.g.q, cp1, p1);
                const double pen = -d2;
                for (int i = 0; i < 3; ++i) { CB(18 + i, j) = n[i]; CB(21 + i, j) = p1[i]; CB(i, j) = p2[i]; }
                CB(24, j) = pen;
                if (!(pen <= W.tol)) bad = 1;
            }
            if (G::any(bad)) {
                if (tid == 0) W.invalid[sc] = 1;
                finish(0);
                break;
            }
            G::sync();
            filter_and_emit<G>(W, S, item, sc, dp, ncon, over, cface, kface, cstate, cb, MC);    // writes pc_count, overflow
            state = ST_DONE;
            break;
        }
        default: state = ST_DONE; break;
        }
    }
    if (state != ST_DONE && round >= last_round) {      // out of rounds: a capacity error, never a silently missing contact
        if (tid == 0) { *pc_count = 0; atomicOr(W.overflow + sc, over | 64); }
        state = ST_DONE;
    }
    G::sync();
    if (tid == 0) {
        hdr[H_STATE] = state; hdr[H_NCAND] = ncand; hdr[H_NCON] = ncon; hdr[H_ITER] = iter; hdr[H_OVER] = over;
        hdr[H_QV] = qv; hdr[H_QG] = qg;
    }
}

__global__ void __launch_bounds__(G::BT) igr_advance_kernel(DssWorld W_arg, int round, int last_round)
{
    DSS_KERNARG_REF(DssWorld, W, W_arg);
    __shared__ ScratchT<G> S;
    int n = W.n_pairs[6];
    if (n > W.igr_items_cap) n = W.igr_items_cap;
    for (int it = blockIdx.x; it < n; it += gridDim.x) {
        advance(W, S, it, round, last_round);
        __syncthreads();
    }
}

}  // namespace

namespace dss {
// All rounds of one detection.  The list lengths of round r live in igr_qn[2 r], igr_qn[2 r + 1] (cleared once up front);
// the point / result buffers alternate between two sets.
int launch_igr_rounds(const DssWorld &W, hipStream_t stream)
{
    if (!W.igr.W0 || !W.igr_hdr || !W.igr_cface || !W.igr_cstate || !W.igr_cbuf || !W.igr_qpts || !W.igr_qlat || !W.igr_qsdf ||
        !W.igr_qgrad || !W.igr_qn || W.igr_items_cap <= 0 || W.igr_qcap <= 0)
        return DSS_E_BADARG;
    const int rounds = W.igr_rounds > 0 ? W.igr_rounds : DSS_IGR_ROUNDS;
    if (rounds > DSS_IGR_ROUNDS) return DSS_E_BADARG;
    (void)hipMemsetAsync(W.igr_qn, 0, (size_t)2 * (DSS_IGR_ROUNDS + 2) * sizeof(int), stream);
    // one workgroup per item; with the caller's expectation (items of the previous detection) no more than that plus slack --
    // the kernel strides over the list, so any grid is correct (see igr_mlp.hip on what an idle workgroup costs)
#if !defined(DSS_IGR_ADV_GRID)
#define DSS_IGR_ADV_GRID 256
#endif
    // (256 VGPRs, 45 KB of LDS: ONE workgroup per CU is resident, so more workgroups than CUs only queue up -- and every one of
    //  them pays its launch with 360 KB of scratch even when its item has long finished)
    int grid = W.igr_items_cap < DSS_IGR_ADV_GRID ? W.igr_items_cap : DSS_IGR_ADV_GRID;
    if (W.igr_hint) { int want = 2 * W.igr_hint[0] + 8; if (want < 64) want = 64; if (want < grid) grid = want; }
    for (int r = 0; r <= rounds; ++r) {
        if (r > 0) {
            // both lists of the round in ONE launch (igr_mlp.hip: igr_query2_kernel)
            const int set = r & 1;
            const size_t ov = (size_t)(2 * set + L_VALUE) * W.igr_qcap, og = (size_t)(2 * set + L_GRAD) * W.igr_qcap;
            if (W.igr_ev) (void)hipEventRecord((hipEvent_t)W.igr_ev[4 * r], stream);
            const int rc = launch_igr_pair(W.igr, W.igr_qpts + ov * 3, W.igr_qlat + ov, W.igr_qn + 2 * r + L_VALUE, W.igr_qsdf + ov,
                                           W.igr_qpts + og * 3, W.igr_qlat + og, W.igr_qn + 2 * r + L_GRAD, W.igr_qsdf + og,
                                           W.igr_qgrad + (size_t)set * W.igr_qcap * 3, W.shape_prm, 3, W.igr_qcap, stream,
                                           W.igr_hint ? W.igr_hint[2 * r + L_VALUE] : -1, W.igr_hint ? W.igr_hint[2 * r + L_GRAD] : -1);
            if (W.igr_ev) (void)hipEventRecord((hipEvent_t)W.igr_ev[4 * r + 1], stream);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(igr_advance_kernel, dim3(grid), dim3(G::BT), 0, stream, W, r, rounds);
    }
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}
}  // namespace dss
