/*
 * diffsdfsim_hip.h -- C ABI of libdiffsdfsim_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the hot path of EmbodiedVision/diffsdfsim.  The
 * reference is pure Python on PyTorch and has no FFI of its own; each entry point below
 * names the reference *Python* interface it replaces (file:line relative to the reference
 * root) and INTEGRATION.md shows the ctypes binding a maintainer would add on that side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (torch allocations);
 *     nothing is allocated, freed or synchronised inside the library; no global state;
 *   - double = IEEE binary64 row-major contiguous, int = int32;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it and the call
 *     returns immediately;
 *   - return value: 0 = enqueued, negative = argument error (see DSS_E_*).  Numerical
 *     per-system outcomes are written to the `status` arrays on the device.
 */
#ifndef DIFFSDFSIM_HIP_H
#define DIFFSDFSIM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSS_ABI_VERSION 1

/* return codes of the launchers */
#define DSS_OK 0
#define DSS_E_BADARG (-1)     /* null pointer / non-positive size */
#define DSS_E_WORKSPACE (-2)  /* workspace smaller than *_workspace_bytes() */
#define DSS_E_UNSUPPORTED (-3)/* size outside the compiled limits */

/* per-system status words written by the LCP kernels (match the reference's exceptions) */
#define DSS_LCP_OK 0
#define DSS_LCP_NOT_SPD 1     /* -> RuntimeError('Q is not SPD.')            lcp.py:109-113  */
#define DSS_LCP_Q_SINGULAR 2  /* -> RuntimeError("...LU factorization on Q") batch.py:417-424 */
#define DSS_LCP_INACCURATE 4  /* best residual > 1 (INACC_ERR condition)      batch.py:165-167 */

int dss_abi_version(void);

/* ------------------------------------------------------------------------------------
 * B1: general dense LCP  --  replaces lcp_physics.lcp.lcp.LCPFunction(...)(Q,p,G,h,A,b,F)
 *     forward : lcp_physics/lcp/lcp.py:48-153 + lcp_physics/lcp/solvers/batch.py:70-231,413-520
 *     backward: lcp_physics/lcp/lcp.py:156-213
 * Shapes: Q[B,nz,nz] p[B,nz] G[B,nineq,nz] h[B,nineq] A[B,neq,nz] b[B,neq] F[B,nineq,nineq].
 * neq may be 0 (A, b ignored).  Termination tests are applied per system.
 * ------------------------------------------------------------------------------------ */
size_t dss_lcp_dense_workspace_bytes(int B, int nz, int nineq, int neq);

int dss_lcp_dense_forward(const double *Q, const double *p, const double *G, const double *h,
                          const double *A, const double *b, const double *F,
                          int B, int nz, int nineq, int neq,
                          double eps, int not_improved_lim, int max_iter, int check_spd,
                          double *zhat, double *lam, double *slack, double *nu,   /* out */
                          int *iters, int *status,                                /* out [B] */
                          void *workspace, size_t workspace_bytes, void *stream);

int dss_lcp_dense_backward(const double *Q, const double *G, const double *A, const double *F,
                           int B, int nz, int nineq, int neq,
                           const double *zhat, const double *lam, const double *slack, const double *nu,
                           const double *dl_dz,
                           double *dQ, double *dp, double *dG, double *dh, double *dA, double *db, double *dF,
                           void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * B2: contact-structured LCP  --  what lcp_physics.physics.engines.PdipmEngine.solve_dynamics
 *     (engines.py:56-82) hands to LCPFunction, without ever forming the dense G and F:
 *       Q  = blockdiag(Mblk[b])  (6x6 per body),  p = pvec,  A/b = equality rows,
 *       contact c between bodies cbody[0][c], cbody[1][c] with unit directions D_0 = n and
 *       D_1..D_ND (ND = fric_dirs/2; the reference appends -D_k, physics3d/world.py:84-94),
 *       contact points p1, p2 (world-frame offsets from the body origins), mu_c, h_c.
 *     Row order inside a contact: [n, +D_1..+D_ND, -D_1..-D_ND, cone]; the matching dense
 *     order of the reference is  n -> c,  k-th friction row -> nc + c*fd + k,  cone -> nc+nc*fd+c.
 *   cop   [B][NF][maxc]  NF = 3*(1+ND)+8 : D_0..D_ND (3 each), p1(3), p2(3), mu, h_n   (SoA)
 *   cbody [B][2][maxc]   int32;   nc [B] int32 (contacts beyond nc[s] are ignored)
 *   lam/slack [B][NR][maxc], NR = fric_dirs+2.   Limits: 6*nb+neq <= 64 (one wavefront).
 *   backward (lcp.py:156-213): dMblk [B][nb][36], dpvec [B][nz], dcop like cop (d/dD, d/dp1,
 *   d/dp2, d/dmu, d/dh_n); dA [B][neq][nz] and db [B][neq] may be NULL.
 * ------------------------------------------------------------------------------------ */
size_t dss_lcp_contact_workspace_bytes(int B, int nb, int neq, int maxc, int fric_dirs);

int dss_lcp_contact_forward(const double *Mblk, const double *pvec, const double *A, const double *bvec,
                            const double *cop, const int *cbody, const int *nc,
                            int B, int nb, int neq, int maxc, int fric_dirs,
                            double eps, int not_improved_lim, int max_iter,
                            double *x, double *lam, double *slack, double *nu, int *iters, int *status,
                            void *workspace, size_t workspace_bytes, void *stream);

int dss_lcp_contact_backward(const double *Mblk, const double *A, const double *cop, const int *cbody,
                             const int *nc, int B, int nb, int neq, int maxc, int fric_dirs,
                             const double *x, const double *lam, const double *slack, const double *nu,
                             const double *dl_dx,
                             double *dMblk, double *dpvec, double *dcop, double *dA, double *db, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFSDFSIM_HIP_H */
