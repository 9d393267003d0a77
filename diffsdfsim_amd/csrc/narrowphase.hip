// narrowphase.hip -- SDF contact detection for the batched stepper (gfx950).
//
// Restates FWContactHandler (sdf_physics/physics3d/contacts.py:217-272) for a batch of scenes:
//   overlap_kernel      _overlap                         contacts.py:27-36
//   narrowphase_kernel  _frank_wolfe                     contacts.py:39-94
//                       _compute_contacts                contacts.py:161-214
//                       _filter_contacts                 contacts.py:97-158
//   compact_kernel      the order in which the callbacks append to world.contacts
// One group of threads owns one (scene, directed body pair a->b) -- a single wavefront when a's mesh is small, a
// 256-thread workgroup for big meshes (see Group below): the triangles of a's mesh
// are searched against b's SDF.  Pair order is canonical (i<j, then i->j before j->i), standing
// in for ODE's unreproducible HashSpace callback order.  Everything that the
// reference decides over the whole candidate set of a pair (early exits of the Frank-Wolfe loop,
// the "all penetrations <= tol" test, greedy normal clustering) is decided over the group.
// Candidate and contact lists are compacted in ascending face order with ballot prefix sums, so
// results are deterministic and cluster seeds match the reference's.
//
// The reference thins each normal cluster with Qhull (3-D hull, falling back to 2-D / 1-D when the
// points are flat).  Here: flatness tests with the same fall-back order, a gift-wrapping hull in
// 2-D (collinear points dropped), min/max in 1-D and a supporting-plane test in 3-D.
// This file is compiled twice: as is (DSS_ALL_SHAPES 0, the lean variant the benchmark configs run: box / sphere / cylinder
// bodies with their analytic meshes) and through narrowphase_all.hip (DSS_ALL_SHAPES 1, the full variant: every primitive,
// and the hull stage prepared for level-set meshes -- coincident contact points, normal clusters beyond the LDS scratch;
// only launch_find_contacts_all is exported).  DssWorld.shape_rare picks the variant at launch.  The lean kernel sits at a
// register-allocation equilibrium: the full variant's extra code costs it 25 %.
#ifndef DSS_ALL_SHAPES
#define DSS_ALL_SHAPES 0
#endif


#include "np_common.h"

#ifndef DSS_NP_WAVES
#define DSS_NP_WAVES 3   // waves per SIMD the register allocator must leave room for (= workgroups per CU)
#endif

namespace {
// ---- _overlap ---------------------------------------------------------------------------------
// "Does any vertex of one body lie in the other's query cube", both ways, for every undirected pair.  One workgroup
// per scene, its four wavefronts take the pairs round robin (no barrier inside a pair: the work per pair is a short
// chain of dependent loads).  The surviving pairs of the scene are appended to the work lists with ONE atomic per
// list and scene: a per-pair atomicAdd on two global counters serialises tens of thousands of wavefronts.
constexpr int OV_RUN = 256;   // vertices per culling box (engine.mesh_table CHUNK)
constexpr int OV_NT = 256;    // eight wavefronts per scene: 28 pairs of an 8-body scene in four rounds
__global__ void __launch_bounds__(OV_NT) overlap_kernel(DssWorld W)
{
    __shared__ unsigned char s_ok[64 * 63 / 2];
    __shared__ unsigned char s_big[64];    // body's mesh is searched by a whole workgroup (list 0) / a wavefront (list 1)
    const int nb = W.nb, nup = nb * (nb - 1) / 2, np = npairs_of(nb);
    const int sc = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (!W.active[sc]) return;
    // (a batch with fewer items than the grid has workgroups -- the time of the launch is then the latency of its longest item:
    // the items that search a big mesh get a whole workgroup, whose four wavefronts scan it a quarter each)
    const bool few = (long)W.B * np <= 256L * DSS_NP_WAVES;
    if (tid < nb) {
        const int nf = W.mesh_nf[W.mesh_id[(size_t)sc * nb + tid]];
        s_big[tid] = nf > WAVE_ITEM_MAX_FACES || (few && nf > 2048);
    }
#if DSS_ALL_SHAPES
    // a pair with a neural SDF body goes to the round-based narrow phase (narrowphase_igr.hip), both directions
    __shared__ unsigned char s_igr[64];
    if (tid < nb) s_igr[tid] = W.shape_type[(size_t)sc * nb + tid] == DSS_SHAPE_IGR;
#endif
    for (int up = wv; up < nup; up += OV_NT / 64) {
        int i = 0, rem = up;
        while (rem >= nb - 1 - i) { rem -= nb - 1 - i; ++i; }
        const int j = i + 1 + rem;
        int ok = !W.no_contact[i * nb + j];
        if (ok) {
            BodyD A, Bd;
            load_body(W, sc, i, A);
            load_body(W, sc, j, Bd);
            for (int dir = 0; dir < 2 && ok; ++dir) {
                const BodyD &src = dir ? Bd : A, &dst = dir ? A : Bd;
                const double s = dst.g.shape.scale;
                Region reg;
                region_of(src.g, dst.g, 1e-9, reg);
                const double *vbox = W.vch_box + (size_t)W.mesh_vch_off[src.mesh] * 6;
                const int nch = (src.nv + OV_RUN - 1) / OV_RUN;
                int found = 0;
                // culling boxes tested 64 at a time; only runs that can reach dst's cube are walked
                for (int cb0 = 0; cb0 < nch && !found; cb0 += 64) {
                    const int ch = cb0 + lane;
                    unsigned long long m = __ballot(ch < nch && box_hits(reg, vbox + (size_t)ch * 6));
                    while (m && !found) {
                        const int b = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        int hit = 0;
#pragma unroll
                        for (int sub = 0; sub < OV_RUN / 64; ++sub) {
                            const int v = (cb0 + b) * OV_RUN + sub * 64 + lane;
                            if (v < src.nv) {
                                double p[3];
                                to_frame(src.g, dst.g, W.verts + (size_t)(src.voff + v) * 3, p);
                                hit |= (-s <= p[0] && p[0] <= s && -s <= p[1] && p[1] <= s && -s <= p[2] && p[2] <= s);
                            }
                        }
                        found = __ballot(hit) != 0ull;
                    }
                }
                ok = found;
            }
        }
        if (lane == 0) {
            s_ok[up] = (unsigned char)ok;
            W.ovl[((size_t)sc * nb + i) * nb + j] = ok;
            if (!ok) {
                W.pc_count[(size_t)sc * np + i * (nb - 1) + (j - 1)] = 0;
                W.pc_count[(size_t)sc * np + j * (nb - 1) + i] = 0;
            }
        }
    }
    __syncthreads();
    // list 0: items a whole workgroup works on (big mesh searched), list 1: one wavefront each
    if (tid == 0) {
        int cnt[2] = {0, 0};
#if DSS_ALL_SHAPES
        int nig = 0;
        for (int up = 0, i = 0, j = 1; up < nup; ++up) {
            if (s_ok[up] && (s_igr[i] || s_igr[j])) nig += 2;
            if (++j == nb) { ++i; j = i + 1; }
        }
        if (nig) {
            int at = atomicAdd(W.n_pairs + 6, nig);
            for (int up = 0, i = 0, j = 1; up < nup; ++up) {
                if (s_ok[up] && (s_igr[i] || s_igr[j])) {
                    s_ok[up] = 0;     // not an item of the analytic lists
                    if (W.igr_list && at + 2 <= W.igr_items_cap) {
                        W.igr_list[at++] = sc * np + i * (nb - 1) + (j - 1);
                        W.igr_list[at++] = sc * np + j * (nb - 1) + i;
                    } else atomicOr(W.overflow + sc, 32);
                }
                if (++j == nb) { ++i; j = i + 1; }
            }
        }
#endif
        for (int up = 0, i = 0, j = 1; up < nup; ++up) {
            if (s_ok[up]) { ++cnt[s_big[i] ? 0 : 1]; ++cnt[s_big[j] ? 0 : 1]; }
            if (++j == nb) { ++i; j = i + 1; }
        }
        const int cap = W.B * np;
        // both list lengths live in one 64-bit word (n_pairs[0], n_pairs[1]): one atomic per scene reserves both ranges
        const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long *>(W.n_pairs),
                                                 (unsigned long long)(unsigned)cnt[0] | ((unsigned long long)(unsigned)cnt[1] << 32));
        int at[2] = {(int)(old & 0xffffffffull), (int)(old >> 32)};
        for (int up = 0, i = 0, j = 1; up < nup; ++up) {
            if (s_ok[up]) {
                const int li = s_big[i] ? 0 : 1, lj = s_big[j] ? 0 : 1;
                W.pair_list[(size_t)li * cap + at[li]++] = sc * np + i * (nb - 1) + (j - 1);
                W.pair_list[(size_t)lj * cap + at[lj]++] = sc * np + j * (nb - 1) + i;
            }
            if (++j == nb) { ++i; j = i + 1; }
        }
    }
}

// ---- the narrow phase -------------------------------------------------------------------------
template <class G> __device__ int narrow_pair(const DssWorld &W, ScratchT<G> &S, int item, int slot_id)
{
#define STAMP(i) do { if (W.dbg_stamps && G::tid() == 0) W.dbg_stamps[(size_t)item * 8 + (i)] = wall_clock64(); } while (0)
    STAMP(0);
    item = dss_uniform(item); slot_id = dss_uniform(slot_id);
    const int np = npairs_of(W.nb);
    const int sc = item / np, dp = item % np, tid = G::tid();
    int a, b;
    pair_of(dp, W.nb, a, b);
    int *pc_count = W.pc_count + (size_t)sc * np + dp;
    BodyD A, Bd;
    load_body(W, sc, a, A);
    load_body(W, sc, b, Bd);
    const int MC = W.max_cand;
    // candidate scratch belongs to the resident group, not to the item: it is reused item after item and stays in L2
    int *__restrict__ cface = W.cand_face + (size_t)slot_id * 2 * MC, *__restrict__ kface = cface + MC;
    int *__restrict__ cstate = W.cand_state + (size_t)slot_id * MC;
    double *__restrict__ cb = W.cand_buf + (size_t)slot_id * DSS_CAND_FIELDS * MC;
    // the mesh table is read-only and disjoint from every scratch array: tell the compiler, so that its loads are
    // not serialised behind the scratch stores
    const double *__restrict__ m_verts = W.verts, *__restrict__ m_fcent = W.fcent;
    const int *__restrict__ m_faces = W.faces;
#define CB(f, k) cb[(size_t)(f) * MC + (k)]
    const double sB = Bd.g.shape.scale;

    // composite transform for the cheap centroid cull (pose-invariant centroids / radii)
    double Ra[9], Rb[9], R12[9], t12[3];
    quat_to_mat(A.g.q, Ra);
    quat_to_mat(Bd.g.q, Rb);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) R12[3 * i + j] = Rb[i] * Ra[j] + Rb[3 + i] * Ra[3 + j] + Rb[6 + i] * Ra[6 + j];
        t12[i] = Rb[i] * (A.g.pos[0] - Bd.g.pos[0]) + Rb[3 + i] * (A.g.pos[1] - Bd.g.pos[1]) + Rb[6 + i] * (A.g.pos[2] - Bd.g.pos[2]);
    }

    // ---- 1. candidate faces (contacts.py:44-52) in ascending face order ------------------------
    // Runs of 256 faces whose culling box misses b's query cube are skipped outright.  Pass A tests the
    // remaining faces and leaves one ballot word per (run, wavefront) in LDS -- no barrier per run;
    // a block scan of the word popcounts gives every candidate its slot; pass B re-derives the triangle
    // of the flagged faces and stores them.  Order = ascending face id, like nonzero(cand_mask).
    int ncand = 0, over = 0;
    Region reg;
    region_of(A.g, Bd.g, 1e-9, reg);
    const double *fbox = W.fch_box + (size_t)W.mesh_fch_off[A.mesh] * 6;
    constexpr int RUN = 256;   // faces per culling box (engine.mesh_table CHUNK)
    const int nch = (A.nf + RUN - 1) / RUN, lane = tid & 63, wv = tid >> 6;
    // cheap pre-test: the face centroid (pose-invariant, composite transform) lies in b's query cube (+ margin)
    auto cull_face = [&](const double *c) -> int {
        double cb2[3];
        for (int i = 0; i < 3; ++i) cb2[i] = R12[3 * i] * c[0] + R12[3 * i + 1] * c[1] + R12[3 * i + 2] * c[2] + t12[i];
        const double lim = sB + 1e-9 * (1.0 + sB);
        return fabs(cb2[0]) <= lim && fabs(cb2[1]) <= lim && fabs(cb2[2]) <= lim;
    };
    // the reference's candidate test (contacts.py:44-52) in its own order of operations
    // (split in two so that the loads of two faces can be in flight before either is tested)
    auto load_tri = [&](int f, double tri[3][3]) {
        const int *fv = m_faces + (size_t)(A.foff + f) * 3;
        for (int k = 0; k < 3; ++k) {
            const double *vp = m_verts + (size_t)(A.voff + fv[k]) * 3;
            for (int i = 0; i < 3; ++i) tri[k][i] = vp[i];
        }
    };
    auto test_tri = [&](double tri[3][3], double pqr[3][3]) -> int {
        double x[3] = {0, 0, 0};
        for (int k = 0; k < 3; ++k) {
            to_frame(A.g, Bd.g, tri[k], pqr[k]);
            for (int i = 0; i < 3; ++i) x[i] += pqr[k][i];
        }
        div3(x, 3.0, x);   // the three exact quotients by one denominator (geom.h)
        double phi, g[3], rad = 0.0;
        // the reference also asks for |grad phi| > 1e-12.  The box gradient is a unit vector wherever the query cube
        // is hit (outside: normalised max(q,0); inside/on the surface: the failsafe direction has norm >= 1) and zero
        // outside the cube, so for boxes that test is the cube test and the gradient need not be evaluated.
        const bool box = Bd.g.shape.type == SHAPE_BOX;
        const bool in_cube = query_sdf(Bd.g.shape, x, phi, g, !box);
        for (int k = 0; k < 3; ++k) {
            const double d[3] = {x[0] - pqr[k][0], x[1] - pqr[k][1], x[2] - pqr[k][2]};
            const double r = t_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            if (r > rad) rad = r;
        }
        if (box) return (phi < rad + W.eps) && in_cube;
        const double gn = t_sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
        return (phi < rad + W.eps) && (gn > 1e-12);
    };
    auto full_face = [&](int f, double pqr[3][3]) -> int {
        double tri[3][3];
        load_tri(f, tri);
        return test_tri(tri, pqr);
    };
    auto test_face = [&](int f, double pqr[3][3]) -> int {
        return cull_face(m_fcent + (size_t)(A.foff + f) * 3) ? full_face(f, pqr) : 0;
    };
    if (G::BT == 64) {
        // one wavefront: (a) centroid pre-test of the runs that can hold a candidate, four independent loads in
        // flight, survivors packed in ascending order into LDS; (b) the full test on dense lanes.
        // A run is also dropped if b's surface is out of reach of every face in it: b's SDF is an exact distance (1-Lipschitz),
        // the run's box holds every face's bounding sphere, so phi_b(centroid) >= phi_b(box centre) - |half diagonal| and
        // rad <= the smallest half extent, and a face with phi_b >= rad + eps is no candidate (contacts.py:52).  A level-set
        // body's whole mesh lies in the query cube of a floor it is nowhere near.
        auto run_far = [&](const double *bx) -> bool {
#if DSS_ALL_SHAPES
            if (Bd.g.shape.type == SHAPE_GRID || Bd.g.shape.type == SHAPE_BOWL || Bd.g.shape.type == SHAPE_IGR) return false;
#endif
            double m[3], e2 = 0.0, emin = INFINITY;
            for (int i = 0; i < 3; ++i) {
                m[i] = 0.5 * (bx[i] + bx[3 + i]);
                const double e = 0.5 * (bx[3 + i] - bx[i]);
                e2 += e * e; emin = fmin(emin, e);
            }
            double pu[3], u, gdum[3];
            for (int i = 0; i < 3; ++i) pu[i] = (R12[3 * i] * m[0] + R12[3 * i + 1] * m[1] + R12[3 * i + 2] * m[2] + t12[i]) / sB;
            sdf_unit(Bd.g.shape, pu, u, gdum, false);
            return u * sB - sqrt(e2) >= emin + W.eps + 1e-9 * (1.0 + sB);
        };
        int npass = 0;
        for (int base = 0; base < nch; base += G::BT) {
            const int ch = base + tid;
            const int hit = ch < nch && box_hits(reg, fbox + (size_t)ch * 6) && !run_far(fbox + (size_t)ch * 6);
            const int slot = compact_slot(hit, npass, S);
            if (slot >= 0 && slot < G::HCAP) S.hidx[slot] = ch;
        }
        if (npass > G::HCAP) return 1;   // more runs in reach than the wavefront's list holds: a workgroup takes the item
        G::sync();
        int *surv = reinterpret_cast<int *>(S.hp);
        constexpr int SCAP = (int)(sizeof(S.hp) / sizeof(int)) - 4 * 64;
        int nsurv = 0;
        auto flush = [&]() {
            G::sync();
            // two tiles of 64 survivors per round: the vertex loads of both are issued before either is tested (every
            // round of this loop is a chain of two dependent global loads, and nothing else hides them)
            for (int base = 0; base < nsurv; base += 128) {
                const int vA = base + lane < nsurv, vB = base + 64 + lane < nsurv;
                const int fA = vA ? surv[base + lane] : 0, fB = vB ? surv[base + 64 + lane] : 0;
                double triA[3][3], triB[3][3], pqr[3][3];
                load_tri(fA, triA);
                load_tri(fB, triB);
                const int flagA = test_tri(triA, pqr) && vA;
                const int slotA = compact_slot(flagA, ncand, S);
                if (slotA >= 0 && slotA < MC) {
                    cface[slotA] = fA;
                    for (int k = 0; k < 3; ++k) for (int i2 = 0; i2 < 3; ++i2) CB(3 * k + i2, slotA) = pqr[k][i2];
                }
                if (base + 64 >= nsurv) break;
                const int flagB = test_tri(triB, pqr) && vB;
                const int slotB = compact_slot(flagB, ncand, S);
                if (slotB >= 0 && slotB < MC) {
                    cface[slotB] = fB;
                    for (int k = 0; k < 3; ++k) for (int i2 = 0; i2 < 3; ++i2) CB(3 * k + i2, slotB) = pqr[k][i2];
                }
            }
            G::sync();
            nsurv = 0;
        };
        for (int i = 0; i < npass; ++i) {
            const int f0 = S.hidx[i] * RUN + lane;
            double c[4][3];
            for (int sub = 0; sub < 4; ++sub) {
                const int f = f0 + 64 * sub < A.nf ? f0 + 64 * sub : A.nf - 1;
                for (int d = 0; d < 3; ++d) c[sub][d] = m_fcent[(size_t)(A.foff + f) * 3 + d];
            }
            for (int sub = 0; sub < 4; ++sub) {
                const int ok = f0 + 64 * sub < A.nf && cull_face(c[sub]);
                const int slot = compact_slot(ok, nsurv, S);
                if (slot >= 0) surv[slot] = f0 + 64 * sub;
            }
            if (nsurv > SCAP) flush();
        }
        if (nsurv > 0) flush();
        if (ncand > MC) { over |= 1; ncand = MC; }
        if (tid == 0) { W.pc_stats[((size_t)sc * np + dp) * 2] = npass; W.pc_stats[((size_t)sc * np + dp) * 2 + 1] = ncand; }
        G::sync();
    } else if (nch <= G::CHCAP) {
        // runs that can hold a candidate: tested in parallel (one culling box per thread), kept in order
        int npass = 0;
        for (int base = 0; base < nch; base += G::BT) {
            const int ch = base + tid;
            const int hit = ch < nch && box_hits(reg, fbox + (size_t)ch * 6);
            const int slot = compact_slot(hit, npass, S);
            if (slot >= 0) S.hidx[slot] = ch;
        }
        unsigned long long *words = reinterpret_cast<unsigned long long *>(S.hp);   // [npass][4]
        G::sync();
        for (int i = 0; i < npass; ++i)
            for (int sub = wv; sub < 4; sub += G::NW) {
                const int f = S.hidx[i] * RUN + sub * 64 + lane;
                double pqr[3][3];
                const int flag = f < A.nf ? test_face(f, pqr) : 0;
                const unsigned long long w = __ballot(flag);
                if (lane == 0) words[i * 4 + sub] = w;
            }
        G::sync();
        // exclusive scan of the popcounts: thread t owns words [t*per, (t+1)*per)
        const int nw = npass * 4, per = (nw + G::BT - 1) / G::BT;
        int mine = 0;
        for (int e = tid * per; e < (tid + 1) * per && e < nw; ++e) mine += __popcll(words[e]);
        S.red_i[tid] = mine;
        G::sync();
        if (tid == 0) { int run = 0; for (int t = 0; t < G::BT; ++t) { const int v = S.red_i[t]; S.red_i[t] = run; run += v; } S.wave_tot[0] = run; }
        G::sync();
        ncand = S.wave_tot[0];
        {
            int run = S.red_i[tid];
            for (int e = tid * per; e < (tid + 1) * per && e < nw; ++e) { S.woff[e] = run; run += __popcll(words[e]); }
        }
        G::sync();
        if (ncand > 0)
            for (int i = 0; i < npass; ++i)
                for (int sub = wv; sub < 4; sub += G::NW) {
                    const unsigned long long w = words[i * 4 + sub];
                    if (!((w >> lane) & 1ull)) continue;
                    const int slot = S.woff[i * 4 + sub] + __popcll(w & ((1ull << lane) - 1ull));
                    if (slot >= MC) continue;
                    const int f = S.hidx[i] * RUN + sub * 64 + lane;
                    double pqr[3][3];
                    test_face(f, pqr);
                    cface[slot] = f;
                    for (int k = 0; k < 3; ++k) for (int i2 = 0; i2 < 3; ++i2) CB(3 * k + i2, slot) = pqr[k][i2];
                }
        if (ncand > MC) { over |= 1; ncand = MC; }
        if (tid == 0) { W.pc_stats[((size_t)sc * np + dp) * 2] = npass; W.pc_stats[((size_t)sc * np + dp) * 2 + 1] = ncand; }
        G::sync();
    } else {
        for (int base = 0; base < A.nf; base += G::BT) {
            if (!box_hits(reg, fbox + (size_t)(base / RUN) * 6)) continue;
            const int f = base + tid;
            double pqr[3][3];
            const int flag = f < A.nf ? test_face(f, pqr) : 0;
            if (!G::any(flag)) continue;
            const int slot = compact_slot(flag, ncand, S);
            if (slot >= 0 && slot < MC) {
                cface[slot] = f;
                for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) CB(3 * k + i, slot) = pqr[k][i];
            }
            if (ncand > MC) { over |= 1; ncand = MC; }
        }
    }
    if (ncand == 0) { if (tid == 0) *pc_count = 0; return 0; }
    G::sync();
    STAMP(1);

    // ---- 2. Frank-Wolfe (contacts.py:57-82) -----------------------------------------------------
    // The first candidate of every thread lives in registers for the whole loop (typical pairs have fewer
    // candidates than threads); further ones stream through the L2-resident scratch.  One barrier per
    // iteration: the two early-exit votes travel as ballots through a parity-double-buffered LDS word.
    struct Cand { double pqr[9], x[3], abc[3]; };
    // vertex bi of the triangle, picked with selects: a run-time index would push the struct into scratch memory
    auto vtx = [](const Cand &c, int bi, int i) { return bi == 0 ? c.pqr[i] : (bi == 1 ? c.pqr[3 + i] : c.pqr[6 + i]); };
    auto load_c = [&](Cand &c, int k) {
        for (int i = 0; i < 9; ++i) c.pqr[i] = CB(i, k);
        for (int i = 0; i < 3; ++i) { c.x[i] = CB(9 + i, k); c.abc[i] = CB(12 + i, k); }
    };
    auto store_c = [&](const Cand &c, int k) {
        for (int i = 0; i < 3; ++i) { CB(9 + i, k) = c.x[i]; CB(12 + i, k) = c.abc[i]; }
    };
    auto init_c = [&](Cand &c) {
        double best = INFINITY; int bi = 0;
        for (int v = 0; v < 3; ++v) {
            double phi, g[3];
            query_sdf(Bd.g.shape, c.pqr + 3 * v, phi, g, false);
            if (phi < best) { best = phi; bi = v; }
        }
        for (int i = 0; i < 3; ++i) { c.x[i] = vtx(c, bi, i); c.abc[i] = (i == bi) ? 1.0 : 0.0; }
    };
    // NOTE the reference forms gamma as python_float * bool_tensor (contacts.py:72-73), which torch
    // promotes to float32: the step sizes, and 1 - gamma, are float32-rounded.  Replicated bit for bit.
    auto eval_c = [&](const Cand &c, int iter, float &gm, int &bi, int &pen) {
        double phi, g[3];
        query_sdf(Bd.g.shape, c.x, phi, g, true);
        double bestd = INFINITY; bi = 0;
        for (int v = 0; v < 3; ++v) {
            const double d = c.pqr[3 * v] * g[0] + c.pqr[3 * v + 1] * g[1] + c.pqr[3 * v + 2] * g[2];
            if (d < bestd) { bestd = d; bi = v; }
        }
        const double impr = (c.x[0] - vtx(c, bi, 0)) * g[0] + (c.x[1] - vtx(c, bi, 1)) * g[1] + (c.x[2] - vtx(c, bi, 2)) * g[2];
        gm = (fabs(impr) > W.tol) ? (float)(2.0 / (iter + 2.0)) : 0.0f;
        pen = phi < -W.tol;
    };
    auto apply_c = [&](Cand &c, float g32, int bi) {
        const double gm = (double)g32, om = (double)(1.0f - g32);
        for (int i = 0; i < 3; ++i) { c.x[i] = om * c.x[i] + gm * vtx(c, bi, i); c.abc[i] *= om; }
        for (int i = 0; i < 3; ++i) if (i == bi) c.abc[i] += gm;
    };
    // iteration 0 for everybody; afterwards only candidates that still move are touched: a candidate whose
    // |improvement| <= tol keeps x, so every later evaluation repeats the same numbers and gamma stays 0
    // (contacts.py:70-73).  The movers (typically a handful on the rim of a flat contact patch) are packed
    // to the front so whole wavefronts drop out of the loop.
    int nmov = 0;
    {
        int vote = 0;
        for (int base = 0; base < ncand; base += G::BT) {
            const int k = base + tid;
            float gm = 0.0f; int bi = 0, pen = 0;
            if (k < ncand) {
                Cand c; load_c(c, k); init_c(c);
                eval_c(c, 0, gm, bi, pen);
                if (gm != 0.0f && !pen) apply_c(c, gm, bi);   // applied only if the loop is not left (decided below)
                store_c(c, k);
                CB(25, k) = (double)gm; cstate[k] = bi;
            }
            const unsigned long long bm = __ballot(gm != 0.0f), bp = __ballot(pen);
            if ((tid & 63) == 0) S.vote[0][tid >> 6] = (bm != 0ull ? 1 : 0) | (bp != 0ull ? 2 : 0);
            G::sync();
            for (int w = 0; w < G::BT / 64; ++w) vote |= S.vote[0][w];
            const int slot = compact_slot(gm != 0.0f, nmov, S);
            if (slot >= 0 && slot < G::HCAP) S.hidx[slot] = k;
        }
        G::sync();
        if (vote & 2) {
            // a penetrating point in iteration 0: the reference leaves the loop BEFORE the update; undo it
            for (int k = tid; k < ncand; k += G::BT) {
                const float gm = (float)CB(25, k);
                if (gm != 0.0f) { Cand c; load_c(c, k); init_c(c); store_c(c, k); }
            }
            nmov = 0;
        }
        if (nmov > G::HCAP) { if (G::BT == 64) return 1; over |= 2; nmov = G::HCAP; }
        G::sync();
    }
    if (G::BT == 64 && nmov > 0 && nmov <= 16) {
        // Few movers, one wavefront: the loop is a serial chain (evaluate -> pick a vertex -> move -> evaluate ...)
        // on a handful of lanes.  The step size of iteration k is known in advance (2 / (k + 2), float32), so the
        // point after the move is one of three: four lanes per mover evaluate the current point (role 0) and the
        // three possible next points (roles 1-3) at once, and every evaluation latency advances TWO iterations.
        // The arithmetic of each iteration is the sequential one, bit for bit.
        const int mi = tid >> 2, role = tid & 3, qbase = tid & ~3;
        Cand m0;
        const int k0 = mi < nmov ? S.hidx[mi] : -1;
        int alive = k0 >= 0;
        if (alive) load_c(m0, k0);
        for (int iter = 1; iter < 32; iter += 2) {
            Cand e = m0;
            if (role) apply_c(e, (float)(2.0 / (iter + 2.0)), role - 1);
            float gm = 0.0f; int bi = 0, pen = 0;
            if (alive) eval_c(e, role ? iter + 1 : iter, gm, bi, pen);
            // iteration `iter`: what role 0 found at the current point
            const float gmA = __shfl(gm, qbase, 64);
            const int biA = __shfl(bi, qbase, 64), penA = __shfl(pen, qbase, 64);
            const unsigned long long bmA = __ballot(alive && gmA != 0.0f), bpA = __ballot(alive && penA);
            if (gmA == 0.0f) alive = 0;                      // froze: x no longer changes
            if (bmA == 0ull || bpA != 0ull) break;           // all gamma == 0, or a penetrating point (contacts.py:74-77)
            if (alive) apply_c(m0, gmA, biA);
            if (iter + 1 >= 32) break;
            // iteration `iter + 1`: what the lane that evaluated the point just moved to found there
            const int src = qbase + 1 + biA;
            const float gmB = __shfl(gm, src, 64);
            const int biB = __shfl(bi, src, 64), penB = __shfl(pen, src, 64);
            const unsigned long long bmB = __ballot(alive && gmB != 0.0f), bpB = __ballot(alive && penB);
            if (gmB == 0.0f) alive = 0;
            if (bmB == 0ull || bpB != 0ull) break;
            if (alive) apply_c(m0, gmB, biB);
        }
        if (k0 >= 0 && role == 0) store_c(m0, k0);
        nmov = 0;     // the general loop below has nothing left to do
    }
    // the mover owned by this thread (if any) stays in registers for the remaining iterations
    Cand m0;
    const int k0 = tid < nmov ? S.hidx[tid] : -1;
    int alive0 = k0 >= 0;
    if (alive0) load_c(m0, k0);
    for (int iter = 1; iter < 32 && nmov > 0; ++iter) {
        float gam[MAX_CPT]; int ind[MAX_CPT];
        int any_pen = 0, moving = 0, q = 1, pen;
        gam[0] = 0.0f; ind[0] = 0;
        if (alive0) {
            eval_c(m0, iter, gam[0], ind[0], pen);
            any_pen |= pen; moving |= gam[0] != 0.0f;
            if (gam[0] == 0.0f) alive0 = 0;     // froze: x no longer changes, every later evaluation repeats this one
        }
        for (int j = tid + G::BT; j < nmov; j += G::BT, ++q) {
            const int k = S.hidx[j];
            gam[q] = 0.0f; ind[q] = 0;
            if (cstate[k] < 0) continue;
            Cand c; load_c(c, k);
            eval_c(c, iter, gam[q], ind[q], pen);
            any_pen |= pen; moving |= gam[q] != 0.0f;
            if (gam[q] == 0.0f) cstate[k] = -1;
        }
        const unsigned long long bm = __ballot(moving), bp = __ballot(any_pen);
        if ((tid & 63) == 0) S.vote[iter & 1][tid >> 6] = (bm != 0ull ? 1 : 0) | (bp != 0ull ? 2 : 0);
        G::sync();
        int vote = 0;
        for (int w = 0; w < G::BT / 64; ++w) vote |= S.vote[iter & 1][w];
        if (!(vote & 1) || (vote & 2)) break;   // all gamma == 0, or a penetrating point (contacts.py:74-77)
        if (gam[0] != 0.0f) apply_c(m0, gam[0], ind[0]);
        q = 1;
        for (int j = tid + G::BT; j < nmov; j += G::BT, ++q) {
            if (gam[q] == 0.0f) continue;
            const int k = S.hidx[j];
            Cand c; load_c(c, k); apply_c(c, gam[q], ind[q]); store_c(c, k);
        }
    }
    if (k0 >= 0) store_c(m0, k0);
    G::sync();

    STAMP(2);
    // ---- 3. pull onto body a's surface, keep phi_b <= eps (contacts.py:84-94) ...
    double qrel[4];
    {
        double qbi[4];
        quat_inv(Bd.g.q, qbi);
        quat_mul(qbi, A.g.q, qrel);
    }
    // ... and 4. the contact geometry of the survivors, in the same round: the triangle is in registers already and the
    // barycentrics need not travel through the scratch (one global round trip per 64 candidates less); compaction in
    // ascending candidate order as before.  The attempt is rejected on penetration (contacts.py:237-240).
    int ncon = 0, bad = 0;
    for (int base = 0; base < ncand; base += G::BT) {
        const int k = base + tid;
        int flag = 0, kf = 0;
        double abc[3] = {0, 0, 0}, n[3] = {0, 0, 0}, p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0}, pen = 0.0;
        if (k < ncand) {
            const int f = cface[k];
            const int *fv = m_faces + (size_t)(A.foff + f) * 3;
            double tri[3][3], xb1[3] = {0, 0, 0};
            for (int v = 0; v < 3; ++v) {
                abc[v] = CB(12 + v, k);
                const double *vp = m_verts + (size_t)(A.voff + fv[v]) * 3;
                for (int i = 0; i < 3; ++i) { tri[v][i] = vp[i]; xb1[i] += tri[v][i] * abc[v]; }
            }
            double phi1, g1[3], gr[3], x[3], phi2, g2[3];
            query_sdf(A.g.shape, xb1, phi1, g1, true);
            quat_apply(qrel, g1, gr);
            for (int i = 0; i < 3; ++i) x[i] = CB(9 + i, k) - phi1 * gr[i];
            query_sdf(Bd.g.shape, x, phi2, g2, false);
            flag = phi2 <= W.eps;
            if (flag) {
                int stable = -1;
                contact_from_bary(A.g, Bd.g, tri, abc, 1e-3, n, p1, p2, pen, &stable);
                kf = stable ? f : (f | DSS_FACE_NORMAL1);     // which body's normal it is travels with the face id
                if (!(pen <= W.tol)) bad = 1;
            }
        }
        if (!G::any(flag)) continue;
        const int slot = compact_slot(flag, ncon, S);
        if (slot >= 0) {
            kface[slot] = kf;
            for (int i = 0; i < 3; ++i) {   // (slot <= k: fields 0-8, pqr, and 15-24 of a slot are dead or unread by now)
                CB(15 + i, slot) = abc[i]; CB(18 + i, slot) = n[i]; CB(21 + i, slot) = p1[i]; CB(i, slot) = p2[i];
            }
            CB(24, slot) = pen;
            // filter state of the contact (np_filter_emit.inc, NP_PRESTATE): 255 = no usable normal
            if (slot < G::HCAP) S.cst[slot] = t_sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) > 1e-12 ? 0 : 255;
        }
    }
    if (ncon == 0) { if (tid == 0) { *pc_count = 0; if (over) atomicOr(W.overflow + sc, over); } return 0; }
    G::sync();

    STAMP(3);
    if (G::any(bad)) {
        if (tid == 0) { W.invalid[sc] = 1; *pc_count = 0; }
        // attempts that will be accepted all the same (decide_kernel: strict_no_penetration=False, dt < dt / 2^10) keep this
        // direction's contacts as they are
        // (dt_try = 0: detection outside a step, World.__init__)
        if (!W.strict_no_pen && W.dt_try[sc] > 0.0 && W.dt_try[sc] < W.dt / 1024.0) { G::sync(); emit_unfiltered<G>(W, sc, dp, ncon, over, kface, cb, MC); }
        return 0;
    }

#define NP_PRESTATE 1
#include "np_filter_emit.inc"
#undef NP_PRESTATE
#undef STAMP
#undef CB
}

// persistent: groups pull items off the active-pair lists through shared cursors so that the long items (a 176 k-face
// floor against a box) do not leave a static-partition tail.  n_pairs = {block-list length, wave-list length, block
// cursor, wave cursor, deferred-list length, deferred cursor}; pair_list = three segments of B*npairs.  Every workgroup first helps with the block list (all four
// wavefronts on one item), then its wavefronts split up and walk the wave list independently.  A wave item that
// outgrows the wave-sized scratch is appended to the deferred list, which a second launch works off block-wise.
union NpScratch {
    ScratchT<BlockGroup> blk;
    ScratchT<WaveGroup> wav[BlockGroup::NW];
};
template <bool DEFERRED> __global__ void __launch_bounds__(NT, DSS_NP_WAVES) narrowphase_kernel(DssWorld W_arg)
{
    DSS_KERNARG_REF(DssWorld, W, W_arg);
    __shared__ NpScratch S;
    __shared__ int s_item;
    const int cap = W.B * npairs_of(W.nb), seg = DEFERRED ? 2 : 0;
    {
        const int n = W.n_pairs[DEFERRED ? 4 : 0];
        const int *list = W.pair_list + (size_t)seg * cap;
        for (;;) {
            __syncthreads();
            if (threadIdx.x == 0) s_item = atomicAdd(W.n_pairs + (DEFERRED ? 5 : 2), 1);
            __syncthreads();
            const int it = s_item;
            if (it >= n) break;
            narrow_pair<BlockGroup>(W, S.blk, list[it], (int)blockIdx.x * BlockGroup::NW);
        }
    }
    if (DEFERRED) return;
    __syncthreads();   // nobody still reads the block scratch
    const int n = W.n_pairs[1], lane = threadIdx.x & 63;
    const int *list = W.pair_list + cap;
    ScratchT<WaveGroup> &Sw = S.wav[threadIdx.x >> 6];
    for (;;) {
        int it = 0;
        if (lane == 0) it = atomicAdd(W.n_pairs + 3, 1);
        it = __shfl(it, 0);
        if (it >= n) break;
        const int item = list[it];
        if (narrow_pair<WaveGroup>(W, Sw, item, (int)(blockIdx.x * BlockGroup::NW + (threadIdx.x >> 6))) && lane == 0)
            W.pair_list[(size_t)2 * cap + atomicAdd(W.n_pairs + 4, 1)] = item;
        dss_wave_sync();
    }
}

// ---- gather the per-pair lists into the scene's contact list in callback order ------------------
// Callback order = undirected pairs (i < j) ascending, i->j before j->i.  The counts of all ordered slots are read
// in one round, a wavefront scan gives their offsets, and every output contact finds its slot by bisection in LDS
// (instead of walking the nb (nb - 1) slots one dependent global load after the other).
__global__ void __launch_bounds__(64) compact_contacts_kernel(DssWorld W, int *nc_out, int *body_out, int *face_out,
                                                               double *abc_out, double *geom_out)
{
    constexpr int MAXSLOT = 64 * 63;
    __shared__ int s_off[MAXSLOT + 1];
    __shared__ short s_a[MAXSLOT], s_b[MAXSLOT];
    __shared__ unsigned char s_neg[MAXSLOT];
    const int sc = blockIdx.x, lane = threadIdx.x, nb = W.nb, np = npairs_of(nb), MP = W.max_pc, MX = W.maxc;
    if (!W.active[sc]) return;
    int run = 0;
    for (int base = 0; base < np; base += 64) {
        const int slot = base + lane;
        int cnt = 0, a = 0, b = 0;
        if (slot < np) {
            int i = 0, rem = slot >> 1;
            while (rem >= nb - 1 - i) { rem -= nb - 1 - i; ++i; }
            const int j = i + 1 + rem;
            a = (slot & 1) ? j : i; b = (slot & 1) ? i : j;
            cnt = W.pc_count[(size_t)sc * np + a * (nb - 1) + (b < a ? b : b - 1)];
        }
        // a negative count = the contacts of a direction that met a penetration in an attempt that goes through anyway
        // (emit_unfiltered): the reference does not search the reverse direction then (contacts.py:237-240)
        const int fwd = __shfl_up(cnt, 1, 64);       // (slots come in pairs i->j, j->i; 64 is even)
        if ((slot & 1) && fwd < 0) cnt = 0;
        if (slot < np) s_neg[slot] = cnt < 0;
        if (cnt < 0) cnt = -cnt;
        int incl = cnt;   // inclusive scan over the wavefront
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
        if (slot < np) { s_off[slot] = run + incl - cnt; s_a[slot] = (short)a; s_b[slot] = (short)b; }
        run += __shfl(incl, 63, 64);
    }
    if (lane == 0) s_off[np] = run;
    __syncthreads();
    const int total = run < MX ? run : MX;
    for (int o = lane; o < total; o += 64) {
        int lo = 0, hi = np;   // last slot with s_off[slot] <= o
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off[mid] <= o) lo = mid; else hi = mid; }
        const int a = s_a[lo], b = s_b[lo], k = o - s_off[lo];
        const size_t dp = (size_t)sc * np + a * (nb - 1) + (b < a ? b : b - 1);
        const int *pf = W.pc_face + dp * MP;
        const double *pabc = W.pc_abc + dp * 3 * MP, *pg = W.pc_geom + dp * 10 * MP;
        // loads first, stores after (no load has to wait behind a possibly aliasing store).  A contact of a penetrating
        // direction was computed under no_grad in the reference: its face id is stored as -1 - face (no geometry adjoint)
        const int face = s_neg[lo] ? -1 - pf[k] : pf[k];
        double abc[3], geo[10];
        for (int f = 0; f < 3; ++f) abc[f] = pabc[(size_t)f * MP + k];
        for (int f = 0; f < 10; ++f) geo[f] = pg[(size_t)f * MP + k];
        body_out[(size_t)sc * 2 * MX + o] = a;
        body_out[(size_t)sc * 2 * MX + MX + o] = b;
        face_out[(size_t)sc * MX + o] = face;
        for (int f = 0; f < 3; ++f) abc_out[((size_t)sc * 3 + f) * MX + o] = abc[f];
        for (int f = 0; f < 10; ++f) geom_out[((size_t)sc * 10 + f) * MX + o] = geo[f];
    }
    if (lane == 0) {
        if (run > MX) atomicOr(W.overflow + sc, 8);
        nc_out[sc] = total;
    }
}

}  // namespace

namespace dss {
// 256 CUs x DSS_NP_WAVES resident workgroups walk the work lists; every wavefront of the grid owns one scratch slot
static inline int np_grid(int B, int nb)
{
    const long items = (long)B * nb * (nb - 1);
    return (int)(items < 256 * DSS_NP_WAVES ? items : 256 * DSS_NP_WAVES);
}
// enqueue detection at the current pose; results land in (nc_out, body_out, ...)
#if DSS_ALL_SHAPES
int launch_igr_rounds(const DssWorld &W, hipStream_t stream);   // narrowphase_igr.hip
int launch_find_contacts_all(const DssWorld &W, int *nc_out, int *body_out, int *face_out, double *abc_out,
                             double *geom_out, hipStream_t stream)
{
#else
int launch_find_contacts_all(const DssWorld &W, int *nc_out, int *body_out, int *face_out, double *abc_out,
                             double *geom_out, hipStream_t stream);
int launch_find_contacts(const DssWorld &W, int *nc_out, int *body_out, int *face_out, double *abc_out,
                         double *geom_out, hipStream_t stream)
{
    if (W.shape_rare) return launch_find_contacts_all(W, nc_out, body_out, face_out, abc_out, geom_out, stream);
#endif
    static_assert(BlockGroup::HCAP <= BlockGroup::BT * MAX_CPT && WaveGroup::HCAP <= WaveGroup::BT * MAX_CPT, "mover registers");
    if (W.max_cand < 1 || W.nb < 1) return DSS_E_UNSUPPORTED;
    if (W.nb < 2) {   // a single body has nothing to collide with: empty contact lists (the no-contact branch, engines.py:40-54)
        hipLaunchKernelGGL(compact_contacts_kernel, dim3(W.B), dim3(64), 0, stream, W, nc_out, body_out, face_out, abc_out, geom_out);
        return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
    }
    const int nup = W.nb * (W.nb - 1) / 2, np = W.nb * (W.nb - 1);
    (void)hipMemsetAsync(W.n_pairs, 0, 8 * sizeof(int), stream);   // counts and cursors of the work lists
    hipLaunchKernelGGL(overlap_kernel, dim3(W.B), dim3(OV_NT), 0, stream, W);
    // 256 CUs x DSS_NP_WAVES resident workgroups walk the work lists; no idle dispatches
    const int grid = np_grid(W.B, W.nb);
    hipLaunchKernelGGL(narrowphase_kernel<false>, dim3(grid), dim3(NT), 0, stream, W);
    // normally finds an empty list: a small grid keeps the empty launch cheap, and works a real list off all the same
    hipLaunchKernelGGL(narrowphase_kernel<true>, dim3(grid < 64 ? grid : 64), dim3(NT), 0, stream, W);
#if DSS_ALL_SHAPES
    if (W.igr_list) { const int rc = launch_igr_rounds(W, stream); if (rc) return rc; }
#endif
    hipLaunchKernelGGL(compact_contacts_kernel, dim3(W.B), dim3(64), 0, stream, W, nc_out, body_out, face_out, abc_out, geom_out);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}
}  // namespace dss

#if !DSS_ALL_SHAPES
extern "C" int dss_np_slots(int B, int nb) { return dss::np_grid(B, nb) * 4; }
#endif
