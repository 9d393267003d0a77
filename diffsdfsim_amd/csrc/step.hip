// step.hip -- the per-attempt pipeline of the batched stepper (gfx950) and its C-ABI launchers.
//
//   assemble_kernel   PdipmEngine.solve_dynamics up to the LCP call   engines.py:31-81
//                     World3D.M / Jc / Jf / mu / restitutions          physics3d/world.py:48-101, world.py:400-501
//   (lcp_contact.hip) the LCP itself
//   integrate_kernel  world.set_v + Body3D.move                        world.py:259-264, bodies.py:488-511
//   (narrowphase.hip) world.find_contacts
//   decide_kernel     accept / halve-and-retry, t += dt                 world.py:270-379
//
// One wavefront per scene for the small per-scene kernels: lane = body or lane = contact, state
// arrays are [scene][...] contiguous so a wave reads one scene's record in a few coalesced lines.
#include <math.h>

#include "../../include/diffsdfsim_hip.h"
#include "contact_geom.h"
#include "wave_utils.h"

namespace dss {
int launch_find_contacts(const DssWorld &W, int *nc_out, int *body_out, int *face_out, double *abc_out,
                         double *geom_out, hipStream_t stream);
}

namespace {
using namespace dss;

__global__ void __launch_bounds__(64) begin_kernel(DssWorld W)
{
    const int sc = blockIdx.x * 64 + threadIdx.x;
    if (sc >= W.B) return;
    if (W.step_mask && !W.step_mask[sc]) { W.active[sc] = 0; return; }      // this scene sits the step out
    const double t = W.t[sc];
    W.t_end[sc] = t + W.dt;          // end_t = self.t + self.dt            (world.py:129)
    W.dt_try[sc] = (t + W.dt) - t;   // dt = end_t - self.t                 (world.py:131)
    W.active[sc] = 1;
    if (W.had_contacts) W.had_contacts[sc] = 0;
    atomicAdd(W.n_active, 1);
}

__global__ void __launch_bounds__(64) assemble_kernel(DssWorld W)
{
    const int sc = blockIdx.x, lane = threadIdx.x, nb = W.nb, MX = W.maxc;
    if (!W.active[sc]) return;
    const int ND = W.fric_dirs / 2, NF = 3 * (1 + ND) + 8;
    // dt_ of this attempt; after a time-of-contact event the reference computes
    // dt_ = -last_dt + (last_dt.detach() + dt_)  (world.py:253-257): same value up to one rounding
    double dt = W.dt_try[sc];
    if (W.toc_diff && W.toc[sc]) { const double l = W.last_dt[sc]; dt = -l + (l + dt); }
    if (lane == 0) { W.dt_use[sc] = dt; W.invalid[sc] = 0; }
    if (lane < nb) {
        const size_t bi = (size_t)sc * nb + lane;
        double ps[7], v[6];
        for (int i = 0; i < 7; ++i) { ps[i] = W.pose[bi * 7 + i]; W.pose0[bi * 7 + i] = ps[i]; }
        for (int i = 0; i < 6; ++i) { v[i] = W.vel[bi * 6 + i]; W.vel0[bi * 6 + i] = v[i]; }
        double Iw[9];
        world_inertia(ps, W.inertia + bi * 9, Iw);
        double *M = W.Mblk + bi * 36;
        const double m = W.mass[bi];
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) M[6 * r + c] = (r < 3 && c < 3) ? Iw[3 * r + c] : ((r == c) ? m : 0.0);
        // u = M v + dt f   (engines.py:36-37)
        for (int r = 0; r < 6; ++r) {
            double acc = 0.0;
            if (r < 3) for (int c = 0; c < 3; ++c) acc += Iw[3 * r + c] * v[c];
            else acc = m * v[r];
            W.pvec[(size_t)sc * 6 * nb + 6 * lane + r] = acc + dt * W.fext[bi * 6 + r];
        }
    }
    const int nc = W.nc[sc];
    double *cop = W.cop + (size_t)sc * NF * MX;
    for (int c = lane; c < nc; c += 64) {
        const int b1 = W.c_body[(size_t)sc * 2 * MX + c], b2 = W.c_body[(size_t)sc * 2 * MX + MX + c];
        W.cop_body[(size_t)sc * 2 * MX + c] = b1;
        W.cop_body[(size_t)sc * 2 * MX + MX + c] = b2;
        const double *g = W.c_geom + (size_t)sc * 10 * MX;
        double n[3], p1[3], p2[3], D[4][3];
        for (int i = 0; i < 3; ++i) { n[i] = g[(size_t)i * MX + c]; p1[i] = g[(size_t)(3 + i) * MX + c]; p2[i] = g[(size_t)(6 + i) * MX + c]; }
        friction_dirs(n, ND, D);
        for (int i = 0; i < 3; ++i) {
            cop[(size_t)i * MX + c] = n[i];
            for (int k = 0; k < ND; ++k) cop[(size_t)(3 * (k + 1) + i) * MX + c] = D[k][i];
        }
        const int o = 3 * (1 + ND);
        for (int i = 0; i < 3; ++i) { cop[(size_t)(o + i) * MX + c] = p1[i]; cop[(size_t)(o + 3 + i) * MX + c] = p2[i]; }
        const size_t i1 = (size_t)sc * nb + b1, i2 = (size_t)sc * nb + b2;
        cop[(size_t)(o + 6) * MX + c] = 0.5 * (W.fric[i1] + W.fric[i2]);       // world.py:478
        // h_n = (Jc v) * restitution  (engines.py:58, world.py:400-407, physics3d/world.py:56-70)
        const double *v1 = W.vel + i1 * 6, *v2 = W.vel + i2 * 6;
        double c1[3], c2[3], jv = 0.0;
        cross(p1, n, c1);
        cross(p2, n, c2);
        for (int i = 0; i < 3; ++i) jv += c1[i] * v1[i] + n[i] * v1[3 + i] - c2[i] * v2[i] - n[i] * v2[3 + i];
        cop[(size_t)(o + 7) * MX + c] = jv * (0.5 * (W.restitution[i1] + W.restitution[i2]));
    }
}

__global__ void __launch_bounds__(64) integrate_kernel(DssWorld W)
{
    const int sc = blockIdx.x, lane = threadIdx.x, nb = W.nb;
    if (!W.active[sc]) return;
    if (lane < nb) {
        const size_t bi = (size_t)sc * nb + lane;
        double v[6], out[7];
        for (int i = 0; i < 6; ++i) { v[i] = -W.x[(size_t)sc * 6 * nb + 6 * lane + i]; W.vel[bi * 6 + i] = v[i]; }  // engines.py:81-82
        const double dt = W.dt_use[sc];
        integrate_pose(W.pose0 + bi * 7, v, dt, out);
        for (int i = 0; i < 7; ++i) W.pose[bi * 7 + i] = out[i];
    }
}

// contacts detected by the attempt (W.n_*), committed to W.c_* on accept
struct NewContacts {
    int *nc, *body, *face;
    double *abc, *geom;
};

__global__ void __launch_bounds__(64) decide_kernel(DssWorld W, NewContacts N)
{
    const int sc = blockIdx.x, lane = threadIdx.x, nb = W.nb, MX = W.maxc;
    if (!W.active[sc]) return;
    // a capacity was exceeded in contact detection (sticky, see DssWorld.overflow): tell the host through the word it
    // reads after every attempt, so that a truncated contact set never goes unnoticed
    if (lane == 0 && W.overflow[sc]) atomicOr(W.n_active, DSS_N_ACTIVE_OVERFLOW);
    const double dt_try = W.dt_try[sc];
    const bool tiny = !W.strict_no_pen && dt_try < W.dt / 1024.0;   // world.py:345-347
    const bool accept = !W.invalid[sc] || tiny;
    if (accept) {
        const int nc_new = N.nc[sc], nc_old = W.nc[sc];
        // toc_contacts: contacts between bodies that had no contact at the start (world.py:272-274).  The body pairs
        // of the old contacts go into an LDS bitmap, which the new contacts then look up.
        __shared__ unsigned s_pairs[64 * 64 / 32];
        for (int w = lane; w < (nb * nb + 31) / 32; w += 64) s_pairs[w] = 0u;
        __syncthreads();
        for (int c = lane; c < nc_old; c += 64) {
            const int bit = W.c_body[(size_t)sc * 2 * MX + c] * nb + W.c_body[(size_t)sc * 2 * MX + MX + c];
            atomicOr(&s_pairs[bit >> 5], 1u << (bit & 31));
        }
        __syncthreads();
        int toc = 0;
        for (int c = lane; c < nc_new; c += 64) {
            const int a = N.body[(size_t)sc * 2 * MX + c], b = N.body[(size_t)sc * 2 * MX + MX + c];
            const int ab = a * nb + b, ba = b * nb + a;
            toc |= !(((s_pairs[ab >> 5] >> (ab & 31)) | (s_pairs[ba >> 5] >> (ba & 31))) & 1u);
        }
        toc = __ballot(toc) != 0ull;
        // the escape of world.py:345-347 leaves the loop before toc_contacts / last_dt are touched
        const bool escaped = W.invalid[sc] != 0;
        if (escaped) toc = W.toc[sc];
        // tape record of the accepted sub-step
        const int slot = W.nsub[sc];
        // the tape is full: the sub-step cannot be recorded, a reverse sweep over it would read past the tape.  A capacity
        // error like the others (sticky, the host raises): max_sub has to be raised.
        if (W.tp_pose && slot >= W.max_sub && lane == 0) { atomicOr(W.overflow + sc, 16); atomicOr(W.n_active, DSS_N_ACTIVE_OVERFLOW); }
        if (W.tp_pose && slot < W.max_sub) {
            const size_t rec = (size_t)slot * W.B + sc;
            if (lane < nb) {
                for (int i = 0; i < 7; ++i) W.tp_pose[(rec * nb + lane) * 7 + i] = W.pose0[((size_t)sc * nb + lane) * 7 + i];
                for (int i = 0; i < 6; ++i) {
                    W.tp_vel[(rec * nb + lane) * 6 + i] = W.vel0[((size_t)sc * nb + lane) * 6 + i];
                    W.tp_x[rec * 6 * nb + 6 * lane + i] = W.x[(size_t)sc * 6 * nb + 6 * lane + i];
                }
            }
            if (lane < W.neq) W.tp_nu[rec * W.neq + lane] = W.nu[(size_t)sc * W.neq + lane];
            const int NR = W.fric_dirs + 2;
            // all loads of a contact first, then all stores: the compiler may not move a load across a store through
            // these (possibly aliasing) pointers, and a load-store-load chain costs one memory round trip per field
            for (int c = lane; c < nc_old; c += 64) {
                const int b1 = W.c_body[(size_t)sc * 2 * MX + c], b2 = W.c_body[(size_t)sc * 2 * MX + MX + c];
                const int face = W.c_face[(size_t)sc * MX + c];
                double abc[3], geo[10], lm[10], sl[10];
                for (int f = 0; f < 3; ++f) abc[f] = W.c_abc[((size_t)sc * 3 + f) * MX + c];
                for (int f = 0; f < 10; ++f) geo[f] = W.c_geom[((size_t)sc * 10 + f) * MX + c];
#pragma unroll
                for (int q = 0; q < 10; ++q)
                    if (q < NR) { lm[q] = W.lam[((size_t)sc * NR + q) * MX + c]; sl[q] = W.slack[((size_t)sc * NR + q) * MX + c]; }
                W.tp_body[rec * 2 * MX + c] = b1;
                W.tp_body[rec * 2 * MX + MX + c] = b2;
                W.tp_face[rec * MX + c] = face;
                for (int f = 0; f < 3; ++f) W.tp_abc[(rec * 3 + f) * MX + c] = abc[f];
                for (int f = 0; f < 10; ++f) W.tp_geom[(rec * 10 + f) * MX + c] = geo[f];
#pragma unroll
                for (int q = 0; q < 10; ++q)
                    if (q < NR) { W.tp_lam[(rec * NR + q) * MX + c] = lm[q]; W.tp_slack[(rec * NR + q) * MX + c] = sl[q]; }
            }
            if (lane == 0) {
                W.tp_dt[rec] = W.dt_use[sc]; W.tp_nc[rec] = nc_old;
                if (W.tp_t) W.tp_t[rec] = W.t[sc];
                W.tp_flags[rec] = ((W.toc_diff && toc && !escaped) ? 1 : 0) | ((W.toc_diff && W.toc[sc]) ? 2 : 0);
            }
        }
        __syncthreads();
        // commit the new contacts
        for (int c = lane; c < nc_new; c += 64) {
            const int b1 = N.body[(size_t)sc * 2 * MX + c], b2 = N.body[(size_t)sc * 2 * MX + MX + c], face = N.face[(size_t)sc * MX + c];
            double abc[3], geo[10];
            for (int f = 0; f < 3; ++f) abc[f] = N.abc[((size_t)sc * 3 + f) * MX + c];
            for (int f = 0; f < 10; ++f) geo[f] = N.geom[((size_t)sc * 10 + f) * MX + c];
            W.c_body[(size_t)sc * 2 * MX + c] = b1;
            W.c_body[(size_t)sc * 2 * MX + MX + c] = b2;
            W.c_face[(size_t)sc * MX + c] = face;
            for (int f = 0; f < 3; ++f) W.c_abc[((size_t)sc * 3 + f) * MX + c] = abc[f];
            for (int f = 0; f < 10; ++f) W.c_geom[((size_t)sc * 10 + f) * MX + c] = geo[f];
        }
        if (lane == 0) {
            W.nc[sc] = nc_new;
            if (W.had_contacts && nc_new > 0) W.had_contacts[sc] = 1;      // `if self.contacts: had_contacts = True` (world.py:133-138)
            W.toc[sc] = toc;
            if (W.toc_diff && toc && !escaped) W.last_dt[sc] = W.dt_use[sc];   // world.py:341
            W.nsub[sc] = slot + 1;
            const double t = W.t[sc] + dt_try;                     // self.t += dt   (world.py:379)
            W.t[sc] = t;
            if (t < W.t_end[sc]) W.dt_try[sc] = W.t_end[sc] - t;   // step(): dt = end_t - self.t
            else if (W.steps_left && W.steps_left[sc] > 1) {
                // the scene's next World.step() begins here (begin_kernel's arithmetic), without waiting for the rest of the batch
                W.steps_left[sc] -= 1;
                W.t_end[sc] = t + W.dt;
                W.dt_try[sc] = (t + W.dt) - t;
                if (W.had_contacts) W.had_contacts[sc] = 0;
            }
            else { W.active[sc] = 0; atomicAdd(W.n_active, -1); }
        }
    } else {
        // dt /= 2, restore p and v; contacts stay those of the sub-step start (world.py:348-356)
        if (lane < nb) {
            const size_t bi = (size_t)sc * nb + lane;
            for (int i = 0; i < 7; ++i) W.pose[bi * 7 + i] = W.pose0[bi * 7 + i];
            for (int i = 0; i < 6; ++i) W.vel[bi * 6 + i] = W.vel0[bi * 6 + i];
        }
        if (lane == 0) W.dt_try[sc] = dt_try / 2.0;
    }
}

inline int check_world(const DssWorld *W)
{
    if (!W) return DSS_E_BADARG;
    if (W->B <= 0 || W->nb <= 0 || W->nb > 64 || W->maxc <= 0) return DSS_E_BADARG;
    if (W->fric_dirs != 4 && W->fric_dirs != 8) return DSS_E_BADARG;
    if (6 * W->nb + W->neq > 64) return DSS_E_UNSUPPORTED;
    return DSS_OK;
}

}  // namespace

extern "C" {

size_t dss_world_sizeof(void) { return sizeof(DssWorld); }

int dss_step_begin(const DssWorld *W, void *stream)
{
    int rc = check_world(W);
    if (rc) return rc;
    (void)hipMemsetAsync(W->n_active, 0, sizeof(int), (hipStream_t)stream);
    hipLaunchKernelGGL(begin_kernel, dim3((W->B + 63) / 64), dim3(64), 0, (hipStream_t)stream, *W);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}

int dss_find_contacts(const DssWorld *W, void *stream)
{
    int rc = check_world(W);
    if (rc) return rc;
    return dss::launch_find_contacts(*W, W->nc, W->c_body, W->c_face, W->c_abc, W->c_geom, (hipStream_t)stream);
}

int dss_solve_dynamics(const DssWorld *W, void *lcp_workspace, size_t lcp_workspace_bytes, void *stream_)
{
    int rc = check_world(W);
    if (rc) return rc;
    hipLaunchKernelGGL(assemble_kernel, dim3(W->B), dim3(64), 0, (hipStream_t)stream_, *W);
    return dss_lcp_contact_forward(W->Mblk, W->pvec, W->Je, W->b_eq, W->cop, W->cop_body, W->nc, W->active, W->B, W->nb,
                                   W->neq, W->maxc, W->fric_dirs, 1e-12, 3, W->lcp_max_iter, W->x, W->lam, W->slack, W->nu,
                                   W->lcp_iters, W->lcp_status, lcp_workspace, lcp_workspace_bytes, stream_);
}

int dss_step_attempt(const DssWorld *W, void *lcp_workspace, size_t lcp_workspace_bytes, void *stream_)
{
    int rc = check_world(W);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(assemble_kernel, dim3(W->B), dim3(64), 0, stream, *W);
    if (W->ev_lcp_start) (void)hipEventRecord((hipEvent_t)W->ev_lcp_start, stream);
    rc = dss_lcp_contact_forward(W->Mblk, W->pvec, W->Je, W->b_eq, W->cop, W->cop_body, W->nc,
                                 W->active, W->B, W->nb, W->neq, W->maxc, W->fric_dirs, 1e-12, 3, W->lcp_max_iter,
                                 W->x, W->lam, W->slack, W->nu, W->lcp_iters, W->lcp_status, lcp_workspace,
                                 lcp_workspace_bytes, stream_);
    if (W->ev_lcp_stop) (void)hipEventRecord((hipEvent_t)W->ev_lcp_stop, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(integrate_kernel, dim3(W->B), dim3(64), 0, stream, *W);
    NewContacts N;
    N.nc = W->n_nc; N.body = W->n_body; N.face = W->n_face; N.abc = W->n_abc; N.geom = W->n_geom;
    if (W->ev_np_start) (void)hipEventRecord((hipEvent_t)W->ev_np_start, stream);
    rc = dss::launch_find_contacts(*W, N.nc, N.body, N.face, N.abc, N.geom, stream);
    if (W->ev_np_stop) (void)hipEventRecord((hipEvent_t)W->ev_np_stop, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(decide_kernel, dim3(W->B), dim3(64), 0, stream, *W, N);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}

}  // extern "C"
