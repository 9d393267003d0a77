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
                            const int *active /* [B] or NULL: scenes with active[s]==0 are skipped */,
                            int B, int nb, int neq, int maxc, int fric_dirs,
                            double eps, int not_improved_lim, int max_iter,
                            double *x, double *lam, double *slack, double *nu, int *iters, int *status,
                            void *workspace, size_t workspace_bytes, void *stream);

int dss_lcp_contact_backward(const double *Mblk, const double *A, const double *cop, const int *cbody,
                             const int *nc, const int *active /* [B] or NULL */,
                             int B, int nb, int neq, int maxc, int fric_dirs,
                             const double *x, const double *lam, const double *slack, const double *nu,
                             const double *dl_dx,
                             double *dMblk, double *dpvec, double *dcop, double *dA, double *db, void *stream);

/* ------------------------------------------------------------------------------------
 * B3: batched world stepper  --  replaces, for a batch of B independent scenes,
 *     World.step / step_dt          lcp_physics/physics/world.py:119-139, 241-379
 *     PdipmEngine.solve_dynamics     lcp_physics/physics/engines.py:31-83   (assembly)
 *     World3D.M/Jc/Jf, mu, restitutions  sdf_physics/physics3d/world.py:48-101, world.py:402-501
 *     Body3D.move / set_p            sdf_physics/physics3d/bodies.py:488-511
 *     FWContactHandler (+_overlap, _frank_wolfe, _compute_contacts, _filter_contacts)
 *                                    sdf_physics/physics3d/contacts.py:27-272
 *     SDF3D.query_sdfs + box/sphere  sdf_physics/physics3d/bodies.py:38-95, 721-760
 * State is struct-of-arrays over the scene axis; all pointers are device pointers owned by
 * the caller.  One "attempt" = solve -> integrate -> detect -> accept/halve for every active
 * scene (the reference's retry loop, world.py:249-356, run in lock step across the batch).
 * ------------------------------------------------------------------------------------ */
/* The nine Linear layers of the IGR network (decode_igr, utils.py:330-350) in the layout the matrix-core kernel reads:
 * W0 [128][5], b0 [128]; Wp = the seven 128x128 layers in MFMA fragment order (dss_igr_packed_doubles() doubles,
 * diffsdfsim_amd/igr.py: pack_weights; layer 3's five missing rows are zero), bh [7][128]; W8 [128], b8 [1]. */
typedef struct DssIgrNet {
    const double *W0, *b0, *Wp, *bh, *W8, *b8;
} DssIgrNet;

typedef struct DssWorld {
    /* sizes */
    int B, nb, neq, maxc, fric_dirs;
    int max_cand;   /* Frank-Wolfe working set per directed body pair            */
    int max_pc;     /* contacts kept per directed body pair after filtering      */
    int nmesh, strict_no_pen, toc_diff /* World3D(time_of_contact_diff=...) */, lcp_max_iter;
    int shape_rare;   /* 0 = lean kernel variants: every shape_type is box / sphere / cylinder and no normal cluster of a
                         body pair exceeds 1024 contacts (analytic meshes); 1 = full variants (all primitives, level-set
                         meshes).  See narrowphase.hip. */
    int grad_flags;   /* World3D's gradient switches (physics3d/world.py:33-37), reverse sweep only: DSS_GRAD_* */
    double eps, tol, dt;   /* Defaults3D.EPSILON / TOL (utils.py:45-48), world dt */
    /* body state [B][nb][7] / [B][nb][6] */
    double *pose, *vel;
    /* body parameters, [B][nb](...) */
    const double *mass, *inertia /*[9]*/, *restitution, *fric, *fext /*[6]*/;
    const int *shape_type;      /* DSS_SHAPE_* */
    const double *shape_prm;    /* [3]: box, rounded box, brick dims | sphere radius | cylinder rad, height | bowl r, d */
    const double *shape_aux;    /* [B][nb]: corner radius r of a rounded box / brick (a constant, not differentiated) */
    const int *mesh_id;
    const unsigned char *no_contact; /* [nb][nb], shared by all scenes */
    /* voxel-grid SDF bodies (DSS_SHAPE_GRID; SDFGrid3D, bodies.py:203-257, 763-775): shape_aux = scale, grid_id [B][nb] = the
       body's grid in the pooled table (-1: none); grid g = doubles grid_data[grid_off[g] ..) of shape grid_dims[g][3], x slowest */
    const int *grid_id, *grid_off, *grid_dims;
    const double *grid_data;
    /* mesh table (body frame) */
    const int *mesh_voff, *mesh_nv, *mesh_foff, *mesh_nf;   /* [nmesh] */
    const double *verts;   /* [NV][3] */
    const int *faces;      /* [NF][3] vertex ids local to the mesh */
    const double *fcent;   /* [NF][3] face centroids, */
    const double *frad;    /* [NF]    max centroid-vertex distance (both pose invariant) */
    const double *vgrad;   /* [NV][3] d vertex / d shape parameter: box d v_k/d dims_k ; sphere d v/d rad ;
                              cylinder (d v_x/d rad, d v_y/d rad, d v_z/d height) */
    /* culling boxes (body frame) of every run of 256 consecutive faces / vertices of a mesh:
       [NCH][6] = lo(3), hi(3); face boxes bound centroid +- radius.  mesh_fch_off/mesh_vch_off [nmesh]. */
    const double *fch_box, *vch_box;
    const int *mesh_fch_off, *mesh_vch_off;
    /* equality rows (joints), constant: Je [B][neq][6 nb], right-hand side b_eq [B][neq] (zeros) */
    const double *Je, *b_eq;
    /* per-scene stepping state [B] */
    double *t, *t_end, *dt_try, *last_dt;
    double *dt_use;  /* dt_ actually integrated in the current attempt (world.py:251-257) */
    int *active;     /* 1 while t < t_end in the current outer step */
    const int *step_mask;  /* optional [B]: scenes with 0 sit the next outer step out (dss_step_begin leaves them inactive): a
                              batch whose scenes are at different times, e.g. after a per-scene undo (optim_sphere.py:163-177) */
    int *had_contacts;  /* [B] any accepted sub-step of the current outer step ended with contacts (World.step's return value,
                           world.py:127-139) */
    int *steps_left;    /* optional [B]: outer steps still to do INCLUDING the current one.  A scene that completes an outer step with
                           steps_left > 1 starts its next one at once (what dss_step_begin does for it: t_end = t + dt, dt = t_end - t)
                           instead of going inactive, so the scenes of a batch run through their `World.step()` calls independently --
                           one that halves its dt at a bounce no longer holds the others for that outer step.  Per-scene results are
                           those of stepping in lock-step, bit for bit (a scene never reads another's state).  NULL: lock-step */
    int *toc;        /* reference's `toc_contacts` non-empty */
    int *nsub;       /* accepted sub-steps so far (tape slot) */
    int *n_active;   /* [1] number of scenes still active after dss_step_decide | DSS_N_ACTIVE_OVERFLOW */
    /* current contacts (geometry at the current pose) [B][...] */
    int *nc;                 /* [B] */
    int *c_body;             /* [B][2][maxc] */
    int *c_face;             /* [B][maxc] face id in the mesh of body 1 (DSS_FACE_ID), | DSS_FACE_NORMAL1 if the contact carries body
                                1's normal (the `stable_mask` decision of contacts.py:184-202, kept for the reverse sweep and for
                                parity tests); stored as -1 - (that word) for a contact without geometry adjoint (world.py:345-347) */
    double *c_abc;           /* [B][3][maxc] barycentrics */
    double *c_geom;          /* [B][10][maxc] n(3) p1(3) p2(3) pen */
    /* contacts detected by the current attempt; committed on accept */
    int *n_nc, *n_body, *n_face;
    double *n_abc, *n_geom;
    /* sub-step start copies (rollback, world.py:344-356) */
    double *pose0, *vel0;
    /* LCP operands / results of the current attempt */
    double *Mblk, *pvec, *cop, *x, *lam, *slack, *nu;
    int *cop_body, *lcp_iters, *lcp_status;
    /* narrow phase scratch */
    int *ovl;                /* [B][nb][nb] overlap flags */
    int *pair_list;          /* [3][B*npairs] active (scene*npairs + directed pair) work items of this attempt:
                                workgroup items, wavefront items, wavefront items deferred to a workgroup */
    int *n_pairs;            /* [8] (index 6 = length of the neural work list igr_list): workgroup-list length, wavefront-list length (one 8-byte aligned pair), their two
                                work cursors, deferred-list length and cursor */
    int *invalid;            /* [B] penetration > tol found in this attempt */
    int *overflow;           /* [B] capacity exceeded, bit mask: 1 max_cand, 2 more than 1024 moving Frank-Wolfe candidates (lean variant: or contacts of one normal cluster) in a pair, 4 max_pc, 8 maxc, 16 max_sub (tape slots), 32 igr_qcap / igr_items_cap, 64 igr_rounds, 128 (lean variant) a contact cluster that needs the exact hull of the full variant */
    int *pc_count;           /* [B][npairs] */
    int *pc_stats;           /* [B][npairs][2] work done for the pair: face runs tested, candidate faces (bench accounting) */
    int *pc_face;            /* [B][npairs][max_pc] */
    double *pc_abc;          /* [B][npairs][3][max_pc] */
    double *pc_geom;         /* [B][npairs][10][max_pc] */
    int *cand_face;          /* [dss_np_slots()][2][max_cand] candidate faces / contact faces (per resident wavefront) */
    int *cand_state;         /* [dss_np_slots()][max_cand] contact list / cluster ids */
    double *cand_buf;        /* [dss_np_slots()][DSS_CAND_FIELDS][max_cand] */
    /* tape for the backward pass: slot-major, [max_sub][B][...] (NULL = do not record) */
    int max_sub;
    double *tp_pose, *tp_vel, *tp_dt, *tp_x, *tp_lam, *tp_slack, *tp_nu, *tp_abc, *tp_geom;
    int *tp_nc, *tp_body, *tp_face;
    int *tp_flags;   /* bit 0: the sub-step ended with a time-of-contact event, bit 1: its dt_ used last_dt (world.py:253-257) */
    double *tp_t;    /* [max_sub][B] world time at the START of the sub-step: the stamp the reference gives the trajectory entry
                        it appends for it (world.py:373-379, before `self.t += dt`); may be NULL */
    /* optional hipEvent_t pair recorded around the LCP launch of dss_step_attempt (bench roofline) */
    void *ev_lcp_start, *ev_lcp_stop;
    void *ev_np_start, *ev_np_stop;   /* same, around the contact-detection launches */
    /* optional [grid of narrowphase][8] phase time stamps (diagnostic runs only; NULL in production) */
    long long *dbg_stamps;
    /* ---- neural SDF bodies (shape_type DSS_SHAPE_IGR; SDF3D(sdf_func=decode_igr(net), params=[latent]), bodies.py:627-760):
       shape_prm[0..1] = the body's latent code, shape_aux = its scale.  A directed pair with a neural body is a work item of
       the round-based narrow phase (narrowphase_igr.hip): items advance from one batch of SDF queries to the next, the
       queries of all items are evaluated together on the matrix cores (igr_mlp.hip).  All NULL / 0 without such bodies. */
    DssIgrNet igr;
    int igr_items_cap;       /* item slots: >= B * (directed pairs with a neural body) */
    int igr_qcap;            /* points per query list */
    int igr_rounds;          /* query rounds per detection (0 = the default that covers every stage) */
    int *igr_list;           /* [igr_items_cap] work items (scene * npairs + directed pair) of this detection */
    int *igr_hdr;            /* [igr_items_cap][DSS_IGR_HDR] state of every item between rounds */
    int *igr_cface;          /* [igr_items_cap][3][max_cand] candidate faces / contact faces / query ranks */
    int *igr_cstate;         /* [igr_items_cap][max_cand] */
    double *igr_cbuf;        /* [igr_items_cap][DSS_CAND_FIELDS][max_cand] */
    double *igr_qpts;        /* [4][igr_qcap][3] query points in the network's unit frame: (buffer set, value | gradient list) */
    int *igr_qlat;           /* [4][igr_qcap] scene * nb + body of the latent code */
    int *igr_qtag;           /* [4][igr_qcap] what the item wants back with the answer (the face id of a candidate test) */
    double *igr_qsdf;        /* [4][igr_qcap] network outputs */
    double *igr_qgrad;       /* [2][igr_qcap][3] d phi / d xyz of the gradient lists */
    int *igr_qn;             /* [2 (DSS_IGR_ROUNDS + 2)] list lengths: value and gradient list of every round */
    const int *igr_hint;     /* optional, HOST memory, [2 (DSS_IGR_ROUNDS + 2)]: what the previous detection found -- [0] the number of
                                neural work items, [2 r + l] the length of list l in round r (r >= 1; the caller copies igr_qn back
                                and writes the item count, n_pairs[6], over entry 0).  Sizes the grids only: every launch strides over
                                whatever the device-side lengths turn out to be.  NULL = grids that fill the chip. */
    void **igr_ev;           /* optional hipEvent_t [4 (DSS_IGR_ROUNDS + 1)] in HOST memory: (start, stop) around the value-list and the
                                gradient-list evaluation of every round (bench roofline); NULL in production */
} DssWorld;

#define DSS_GRAD_STOP_CONTACT 1    /* stop_contact_grad: Jc (and h = Jc v) from detached contact geometry, world.py:59-62 */
#define DSS_GRAD_STOP_FRICTION 2   /* stop_friction_grad: Jf from detached contact geometry, world.py:77-80 */
#define DSS_GRAD_DETACH_B2 4       /* detach_contact_b2: the contact point in body 2's frame is a constant, contacts.py:175-178 */
#define DSS_FACE_NORMAL1 (1 << 30)        /* flag in a contact's face word: normal = -R1 n1 (body 1's), not R2 n2 */
#define DSS_FACE_ID(w) ((w) & (DSS_FACE_NORMAL1 - 1))
#define DSS_N_ACTIVE_OVERFLOW (1 << 30)   /* set in n_active[0] once any scene's overflow word is non-zero */
#define DSS_CAND_FIELDS 28  /* pqr(9) x(3) abc(3) | abc_k(3) n(3) p1(3) pen spare(3) */
#define DSS_CSCR_ROWS 56
#define DSS_SHAPE_BOX 0
#define DSS_SHAPE_SPHERE 1
#define DSS_SHAPE_CYLINDER 2   /* shape_prm = (rad, height, -), axis = body z */
#define DSS_SHAPE_BOX_ROUNDED 3 /* SDFBoxRounded (bodies.py:857-870): shape_prm = outer dims, shape_aux = r */
#define DSS_SHAPE_BRICK 4      /* SDFBrick (bodies.py:873-885): shape_prm = dims, shape_aux = r (x-y corners rounded) */
#define DSS_SHAPE_BOWL 5       /* SDFBowl (bodies.py:1013-1027): shape_prm = (r, d, -), opening towards +z */
#define DSS_SHAPE_GRID 7       /* SDFGrid3D (bodies.py:763-775): samples of the SDF over the body's unit cube, shape_aux = scale */
#define DSS_SHAPE_IGR 6        /* SDF3D with decode_igr (bodies.py:627-760, utils.py:330-350): shape_prm = latent code (2), shape_aux = scale */
#define DSS_IGR_HDR 16         /* ints of per-item state of the round-based narrow phase */
#define DSS_IGR_ROUNDS 42      /* query rounds that cover every stage: candidates 2, Frank-Wolfe 1 + 31, projection 2, geometry 4, spare */

size_t dss_world_sizeof(void);
/* scratch slots the narrow phase needs for a batch of B scenes with nb bodies (sizes cand_face/cand_state/cand_buf) */
int dss_np_slots(int B, int nb);   /* sizeof(DssWorld): lets a binding check its mirror struct */

/* Start an outer step of length W->dt for every scene: t_end = t + dt, active = 1 (world.py:119-134). */
int dss_step_begin(const DssWorld *W, void *stream);
/* One attempt for all active scenes.  Enqueues: assemble -> LCP -> integrate -> detect -> decide.
 * After it completes W->n_active[0] holds the number of scenes that still have t < t_end.      */
int dss_step_attempt(const DssWorld *W, void *lcp_workspace, size_t lcp_workspace_bytes, void *stream);
/* Engine plug-in (boundary B2), PdipmEngine.solve_dynamics(world, dt) (engines.py:31-83) for every scene with
 * active[s] != 0: assembles from the current state / contacts with dt = dt_try[s] and solves; the new velocities
 * are -W->x.  Nothing else is modified (no integration, no detection). */
int dss_solve_dynamics(const DssWorld *W, void *lcp_workspace, size_t lcp_workspace_bytes, void *stream);
/* Contact detection only, at the current pose (World.__init__, world.py:96-100). */
int dss_find_contacts(const DssWorld *W, void *stream);

/* ------------------------------------------------------------------------------------
 * Backward of the batched stepper: reverse sweep over the tape written by dss_step_attempt
 * (one record per accepted sub-step).  Replaces what torch.autograd does in the reference for
 *   LCPFunctionFn.backward                      lcp_physics/lcp/lcp.py:156-213
 *   the graph of Jc/Jf/M/u assembly              engines.py:36-81, physics3d/world.py:48-101
 *   Body3D.move / set_p                          physics3d/bodies.py:488-511
 *   _compute_contacts on the filtered contacts   physics3d/contacts.py:161-214, 262-264
 * a_pose / a_vel: on entry d(loss)/d(state after the newest unprocessed sub-step), on exit
 * d(loss)/d(state before the oldest processed one).  a_geom carries the adjoint of the contact
 * geometry across calls.  g_* accumulate d(loss)/d(parameters) ([B][nb](...), same shapes as the
 * world's parameter arrays).  cur_slot[s] is the next tape slot to process for scene s, a call
 * processes it iff cur_slot[s] >= lo_slot[s], then decrements it.
 * ------------------------------------------------------------------------------------ */
typedef struct DssAdjoint {
    double *a_pose, *a_vel, *a_geom;
    double *a_last_dt;       /* [B] adjoint of World.last_dt carried to the sub-step that produced it */
    double *a_dt;            /* [B] scratch: adjoint of dt_ of the sub-step being processed */
    double *g_mass, *g_inertia, *g_rest, *g_fric, *g_fext, *g_prm;
    double *g_verts;         /* [NV][3] adjoint of the mesh table's vertices (the contact point is a barycentric combination of
                                a triangle's vertices, contacts.py:165-171); accumulated with atomics by the full kernel
                                variant only (DssWorld.shape_rare), so that level-set meshes carry shape gradients; may be NULL */
    int *cur_slot, *lo_slot;
    /* scratch */
    int *bw_active;          /* [B] */
    double *a_x;             /* [B][6 nb] */
    double *dMblk, *dpvec, *dcop;   /* LCP backward outputs, shapes of Mblk / pvec / cop */
    double *cscr;            /* [B][DSS_CSCR_ROWS][maxc] per-contact VJP pieces */
    int *bw_nc;              /* [B] */
    /* neural SDF bodies: the network is re-evaluated at the contact points of the sub-step being undone (phi, d phi / d xyz
       and d phi / d latent: what the reference's autograd keeps of SDF3D.query_sdfs, bodies.py:730-745 -- the value carries
       the graph, the normal does not).  One compacted point list per dss_step_backward call; NULL without such bodies. */
    int *igr_bw_n;           /* [1] points in the list */
    int *igr_bw_idx;         /* [B][2][maxc] list slot of (contact, body 1 | body 2), -1 = that body is analytic */
    double *igr_bw_pts;      /* [B 2 maxc][3] */
    int *igr_bw_lat;         /* [B 2 maxc] */
    double *igr_bw_sdf;      /* [2][B 2 maxc] phi from the xyz pass / the latent pass */
    double *igr_bw_grad;     /* [2][B 2 maxc][3] d phi / d xyz, d phi / d latent */
} DssAdjoint;

size_t dss_adjoint_sizeof(void);
int dss_step_backward(const DssWorld *W, const DssAdjoint *A, void *stream);

/* ------------------------------------------------------------------------------------
 * IGR neural SDF  --  decode_igr + the autograd input gradient of SDF3D.query_sdfs
 *     sdf_physics/physics3d/utils.py:330-350, sdf_physics/physics3d/bodies.py:730-745
 * Network of IGR_data/train_configs/bob_spot_setup.conf:38-45 (5 -> 128 x3 -> 123, skip, 128 x4 -> 1,
 * Softplus(beta=100)), float64, on the fp64 matrix cores.  pts [n][3] body-frame points (already divided by the
 * body scale), latent [2]; W0 [128][5], b0 [128]; Wp = the seven 128x128 layers in MFMA fragment order
 * (dss_igr_packed_doubles() doubles, see diffsdfsim_amd/igr.py: pack_weights; layer 3's five missing rows are
 * zero), bh [7][128]; W8 [128], b8 [1].  Outputs sdf [n] and d sdf / d xyz [n][3] (not normalised).
 * ------------------------------------------------------------------------------------ */
size_t dss_igr_packed_doubles(void);
/* what a query round evaluates per point */
#define DSS_IGR_XYZ 0      /* phi and d phi / d xyz (SDF3D.query_sdfs with return_grads, bodies.py:730-745) */
#define DSS_IGR_LATENT 1   /* phi and d phi / d latent (MeshSDF backward, bodies.py:680-702; the stepper's latent adjoint) */
#define DSS_IGR_VALUE 2    /* phi only (return_grads=False: candidate test, Laplacian probes, contacts.py:46-60, 184-196) */
/* One evaluation round over a point list: pts [n][3] in the network's unit frame, point i uses the latent code
 * latents[lat_idx[i] * lat_stride + 0..1] (lat_idx NULL: code 0).  The list length is *n_dev if n_dev is non-NULL (device
 * memory, at most n_cap), else n_cap.  sdf [n]; grad [n][3] (unused for DSS_IGR_VALUE). */
int dss_igr_query_list(const DssIgrNet *net, const double *pts, const int *lat_idx, const double *latents, int lat_stride,
                       const int *n_dev, int n_cap, int mode, double *sdf, double *grad, void *stream);
int dss_igr_query(const double *pts, const double *latent, const double *W0, const double *b0, const double *Wp,
                  const double *bh, const double *W8, const double *b8, int n, double *sdf, double *grad, void *stream);
/* Same evaluation, tangents on the latent code: grad [n][3] = (d sdf / d latent_0, d sdf / d latent_1, 0). */
int dss_igr_query_latent_grad(const double *pts, const double *latent, const double *W0, const double *b0, const double *Wp,
                              const double *bh, const double *W8, const double *b8, int n, double *sdf, double *grad, void *stream);

/* ------------------------------------------------------------------------------------
 * World-construction side of the path
 *
 * dss_sdf_query replaces SDF3D.query_sdfs (sdf_physics/physics3d/bodies.py:721-760) for the analytic primitives:
 *   pts [n][3] in the body frame (world units), prm [4] (host) = shape parameters in world units (box, rounded box,
 *   brick: dims; sphere: rad; cylinder: rad, height; bowl: r, d) and, fourth, the corner radius of a rounded box / brick
 *   (bodies.py:38-185, 778-1027) -> sdf [n], grad [n][3] (normalised; may be NULL = return_grads False),
 *   overlap_mask [n] (1 inside the [-scale, scale]^3 query cube; may be NULL).  Outside the cube sdf = scale, grad = 0.
 * dss_mesh_inertia replaces get_ang_inertia (bodies.py:260-395) for a pooled table of closed triangle meshes
 *   (same table layout as DssWorld: verts [NV][3], faces [NF][3] mesh-local indices, mesh_voff/foff/nf [nmesh]):
 *   J [nmesh][9] inertia about the mesh origin for mass[m] at uniform density, volume [nmesh] (may be NULL).
 * ------------------------------------------------------------------------------------ */
int dss_sdf_query(int shape_type, const double *prm, const double *pts, int n, double *sdf, double *grad,
                  unsigned char *overlap_mask, void *stream);
int dss_mesh_inertia(const double *verts, const int *faces, const int *mesh_voff, const int *mesh_foff, const int *mesh_nf,
                     int nmesh, const double *mass, double *J, double *volume, void *stream);
/* The same query for SDFGrid3D (bodies.py:203-241, 763-775): grid [n0][n1][n2] (device) = SDF samples over the body's unit
 * cube, scale = the body's scale.  Value by trilinear interpolation, gradient from the interpolated central-difference
 * field, normalised.  (`grid_interp` of the un-vendored ev_sdf_utils is restated as plain trilinear interpolation.) */
int dss_grid_sdf_query(const double *grid, int n0, int n1, int n2, double scale, const double *pts, int n, double *sdf,
                       double *grad, unsigned char *overlap_mask, void *stream);

/* Adjoint of dss_mesh_inertia for ONE mesh: grad_verts [nv][3] = d (sum_ab grad_J[a][b] J[a][b]) / d verts
 * (the reference differentiates get_ang_inertia with autograd, bodies.py:380-395). */
int dss_mesh_inertia_backward(const double *verts, const int *faces, int nv, int nf, double mass, const double *grad_J,
                              double *grad_verts, void *stream);

/* Self-test of the shared-reciprocal triple division the geometry kernels use (csrc/geom.h: div3): counts the
 * quotients num[i][k] / den[i] whose bit pattern differs from an IEEE division on the device. */
int dss_selftest_div3(const double *num, const double *den, int n, int *mismatches, void *stream);
/* The same for the unscaled square root (csrc/geom.h: t_sqrt) against sqrt() on the device. */
int dss_selftest_sqrt(const double *x, int n, int *mismatches, void *stream);

/* Marching cubes + MeshSDF backward: replaces SDF3D._diff_marching_cubes (bodies.py:653-704).
 *   phi [n0][n1][n2] SDF samples (x slowest, as torch.meshgrid(...).reshape(res,res,res)); inside <=> phi < iso.
 *   ntri_tab [256], tri_tab [256][max_tri][3] (cube-edge ids): diffsdfsim_amd/mc_tables.py.
 *   dss_mc_count  -> totals[0] = vertices, totals[1] = triangles (device ints; read them, allocate, then)
 *   dss_mc_emit   -> verts [V][3] in grid-index units (the caller maps to [-1,1]: v/(res-1)*2-1), faces [F][3].
 *   The workspace passed to dss_mc_emit must be the one dss_mc_count filled.
 *   dss_meshsdf_backward: grad_prm[4] = sum_v -(grad_verts[v] . n_v) d phi / d unit_prm (v), analytic primitives,
 *   unit frame of the reference (unit_prm [4] host = the three parameters and the corner radius, already divided by the
 *   body scale; vertices in [-1,1]^3). */
size_t dss_mc_workspace_bytes(int n0, int n1, int n2);
int dss_mc_count(const double *phi, int n0, int n1, int n2, double iso, const int *ntri_tab, void *workspace,
                 size_t workspace_bytes, int *totals, void *stream);
int dss_mc_emit(const double *phi, int n0, int n1, int n2, double iso, const int *ntri_tab, const signed char *tri_tab,
                int max_tri, const void *workspace, double *verts, int *faces, void *stream);
int dss_meshsdf_backward(int shape_type, const double *unit_prm, const double *unit_verts, const double *grad_verts, int nv,
                         double *grad_prm, void *stream);

/* ------------------------------------------------------------------------------------
 * R18: analytic 2-D contacts -- replaces DiffContactHandler.__call__ (lcp_physics/physics/contacts.py:55-215; the
 * reference's 2-D world of BASELINE configs[0]) for a batch of body pairs, with its vector-Jacobian product (the reference
 * gets that from autograd through the same arithmetic).  A body is a circle (kind 0) or a convex polygon (kind 1, nv <=
 * DSS_C2D_MAXV vertices about its centroid in world orientation, clockwise in the reference's y-down frame: bodies.py
 * Hull.verts).  Arrays are [2][npairs]... : body 1 of every pair, then body 2.
 *   kind, nv [2][P]; pos [2][P][2]; rad [2][P] (circles); verts [2][P][maxv][2];
 *   sat_in / sat_out [2][P]: Hull.last_sat_idx, the edge the separating-axis loops start from (state of the body:
 *     contacts.py:124-131, 153-158, 236) before / after the call;
 *   count [P] in {0, 1, 2}; out [P][2][7] = normal (2, from body 2 to body 1), p1 (2, offset from body 1's position),
 *     p2 (2, from body 2's), penetration -- the tuple the reference appends to world.contacts (contacts.py:208-209).
 *   backward: gout [P][2][7] (rows >= count ignored) -> g_pos [2][P][2], g_rad [2][P], g_verts [2][P][maxv][2], with the
 *     sat_in the forward started from (the branch decisions are taken on values, as the reference takes them on .item()). */
#define DSS_C2D_MAXV 8
/* (a pair whose kind is neither 0 nor 1, or whose vertex count is negative or exceeds maxv / the compiled table of 8, is reported
 *  with count = -1 and writes no contact; the backward pass clamps such counts and returns zeros for what it cannot read) */
int dss_contacts2d_forward(int npairs, int maxv, const int *kind, const int *nv, const double *pos, const double *rad,
                           const double *verts, const int *sat_in, double eps, int *sat_out, int *count, double *out,
                           void *stream);
int dss_contacts2d_backward(int npairs, int maxv, const int *kind, const int *nv, const double *pos, const double *rad,
                            const double *verts, const int *sat_in, double eps, const double *gout, double *g_pos,
                            double *g_rad, double *g_verts, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFSDFSIM_HIP_H */
