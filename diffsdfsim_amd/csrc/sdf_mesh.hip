// sdf_mesh.hip -- SDF point queries and polyhedral mass properties for the world-construction side of the path.
//
//   dss_sdf_query      SDF3D.query_sdfs (sdf_physics/physics3d/bodies.py:721-760) for the analytic primitives
//                      (box / sphere / cylinder, bodies.py:38-124): phi, normalised gradient, overlap mask.
//   dss_mesh_inertia   get_ang_inertia (bodies.py:260-395): Mirtich / volInt.c volume integrals of a closed triangle
//                      mesh -> inertia tensor about the origin for a given mass.
//
// Queries: one lane per point, fully coalesced ([n][3] in, [n] / [n][3] out), HBM-bound by construction
// (24 B in, 33 B out per point against ~250 fp64 instructions: the kernel is still VALU-bound on div/sqrt).
// Inertia: one 256-thread workgroup per mesh; every thread walks faces f = tid, tid + 256, ... and keeps ten partial
// sums (T0, T1[3], T2[3], TP[3]); the partials are combined by a fixed LDS tree, so the result is reproducible.
#include <math.h>

#include "../../include/diffsdfsim_hip.h"
#include "geom.h"

namespace {
using namespace dss;

__global__ void __launch_bounds__(256) sdf_query_kernel(int type, double p0, double p1, double p2, double aux, const double *pts, int n,
                                                       double *sdf, double *grad, unsigned char *mask)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Shape<double> s;
    const double prm[3] = {p0, p1, p2};
    make_shape(s, type, prm, aux);
    const double pt[3] = {pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]};
    double phi, g[3] = {0.0, 0.0, 0.0};
    const bool in = query_sdf(s, pt, phi, g, grad != nullptr);
    sdf[i] = phi;
    if (grad) { grad[3 * (size_t)i] = g[0]; grad[3 * (size_t)i + 1] = g[1]; grad[3 * (size_t)i + 2] = g[2]; }
    if (mask) mask[i] = in ? 1 : 0;
}

// SDFGrid3D.query_sdfs (bodies.py:203-241, 721-775): the body's SDF is a voxel grid over its unit cube.  Value: trilinear
// interpolation at inds = (p + 1) / 2 * (n - 1); gradient: the central-difference field (zero in the boundary layers) of
// the grid, interpolated the same way, normalised (grid_sdf_grad) and normalised again (query_sdfs).  The interpolation
// itself is `ev_sdf_utils.grid_interp`, un-vendored: trilinear with the cell index clamped to [0, n - 2].
__global__ void __launch_bounds__(256) grid_sdf_query_kernel(const double *__restrict__ G, int n0, int n1, int n2, double scale,
                                                            const double *pts, int n, double *sdf, double *grad, unsigned char *mask)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const double pt[3] = {pts[3 * (size_t)t], pts[3 * (size_t)t + 1], pts[3 * (size_t)t + 2]};
    const bool in = fabs(pt[0]) <= scale && fabs(pt[1]) <= scale && fabs(pt[2]) <= scale;
    if (mask) mask[t] = in ? 1 : 0;
    if (!in) {
        sdf[t] = scale;
        if (grad) { grad[3 * (size_t)t] = 0.0; grad[3 * (size_t)t + 1] = 0.0; grad[3 * (size_t)t + 2] = 0.0; }
        return;
    }
    const int nn[3] = {n0, n1, n2};
    int i0[3];
    double w[3];
    for (int d = 0; d < 3; ++d) {
        const double ind = (pt[d] / scale + 1.0) * 0.5 * (double)(nn[d] - 1);
        int c = (int)floor(ind);
        c = c < 0 ? 0 : (c > nn[d] - 2 ? nn[d] - 2 : c);
        i0[d] = c; w[d] = ind - (double)c;
    }
    auto at = [&](int i, int j, int k) { return G[((size_t)i * n1 + j) * n2 + k]; };
    double phi = 0.0, g[3] = {0.0, 0.0, 0.0};
    for (int dx = 0; dx < 2; ++dx)
        for (int dy = 0; dy < 2; ++dy)
            for (int dz = 0; dz < 2; ++dz) {
                const double wt = (dx ? w[0] : 1.0 - w[0]) * (dy ? w[1] : 1.0 - w[1]) * (dz ? w[2] : 1.0 - w[2]);
                const int i = i0[0] + dx, j = i0[1] + dy, k = i0[2] + dz;
                phi = phi + at(i, j, k) * wt;
                if (grad) {
                    const double cx = (i == 0 || i == n0 - 1) ? 0.0 : (at(i + 1, j, k) - at(i - 1, j, k)) / 2.0;
                    const double cy = (j == 0 || j == n1 - 1) ? 0.0 : (at(i, j + 1, k) - at(i, j - 1, k)) / 2.0;
                    const double cz = (k == 0 || k == n2 - 1) ? 0.0 : (at(i, j, k + 1) - at(i, j, k - 1)) / 2.0;
                    g[0] = g[0] + cx * wt; g[1] = g[1] + cy * wt; g[2] = g[2] + cz * wt;
                }
            }
    sdf[t] = phi * scale;
    if (grad) {
        double g1[3], g2[3];
        normalize(g, g1);
        normalize(g1, g2);
        grad[3 * (size_t)t] = g2[0]; grad[3 * (size_t)t + 1] = g2[1]; grad[3 * (size_t)t + 2] = g2[2];
    }
}

// The ten volume-integral contributions of one face (comp_projection_integrals / comp_face_integrals /
// comp_volume_integrals, bodies.py:260-377): c[0] -> T0, c[1..3] -> 2 T1, c[4..6] -> 3 T2, c[7..9] -> 2 TP.
// Templated on the scalar so that the backward differentiates the same code with dual numbers.
template <class T> __device__ inline T pick3(int ax, int A, int B, const T &a, const T &b, const T &c) { return ax == A ? a : (ax == B ? b : c); }
template <class T> __device__ inline void face_integrals(const T v[3][3], T *c)
{
    T nrm[3];
    const T e1[3] = {v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2]};
    const T e2[3] = {v[2][0] - v[1][0], v[2][1] - v[1][1], v[2][2] - v[1][2]};
    cross(e1, e2, nrm);
    const T ln = t_sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
    for (int i = 0; i < 3; ++i) nrm[i] = nrm[i] / ln;
    const T w = -(nrm[0] * v[0][0] + nrm[1] * v[0][1] + nrm[2] * v[0][2]);
    int C = 0;   // torch.argmax: first index of the maximum
    for (int i = 1; i < 3; ++i) if (fabs(val(nrm[i])) > fabs(val(nrm[C]))) C = i;
    const int A = (C + 1) % 3, B = (A + 1) % 3;
    T P1(0.0), Pa(0.0), Paa(0.0), Paaa(0.0), Pb(0.0), Pbb(0.0), Pbbb(0.0), Pab(0.0), Paab(0.0), Pabb(0.0);
    for (int e = 0; e < 3; ++e) {
        const T a0 = pick3(A, 0, 1, v[e][0], v[e][1], v[e][2]), b0 = pick3(B, 0, 1, v[e][0], v[e][1], v[e][2]);
        const T a1 = pick3(A, 0, 1, v[(e + 1) % 3][0], v[(e + 1) % 3][1], v[(e + 1) % 3][2]);
        const T b1 = pick3(B, 0, 1, v[(e + 1) % 3][0], v[(e + 1) % 3][1], v[(e + 1) % 3][2]);
        const T da = a1 - a0, db = b1 - b0;
        const T a0_2 = a0 * a0, a0_3 = a0_2 * a0, a0_4 = a0_3 * a0, b0_2 = b0 * b0, b0_3 = b0_2 * b0, b0_4 = b0_3 * b0;
        const T a1_2 = a1 * a1, a1_3 = a1_2 * a1, b1_2 = b1 * b1, b1_3 = b1_2 * b1;
        const T C1 = a1 + a0, Ca = a1 * C1 + a0_2, Caa = a1 * Ca + a0_3, Caaa = a1 * Caa + a0_4;
        const T Cb = b1 * (b1 + b0) + b0_2, Cbb = b1 * Cb + b0_3, Cbbb = b1 * Cbb + b0_4;
        const T Cab = 3.0 * a1_2 + 2.0 * a1 * a0 + a0_2, Kab = a1_2 + 2.0 * a1 * a0 + 3.0 * a0_2;
        const T Caab = a0 * Cab + 4.0 * a1_3, Kaab = a1 * Kab + 4.0 * a0_3;
        const T Cabb = 4.0 * b1_3 + 3.0 * b1_2 * b0 + 2.0 * b1 * b0_2 + b0_3, Kabb = b1_3 + 2.0 * b1_2 * b0 + 3.0 * b1 * b0_2 + 4.0 * b0_3;
        P1 = P1 + db * C1; Pa = Pa + db * Ca; Paa = Paa + db * Caa; Paaa = Paaa + db * Caaa;
        Pb = Pb + da * Cb; Pbb = Pbb + da * Cbb; Pbbb = Pbbb + da * Cbbb;
        Pab = Pab + db * (b1 * Cab + b0 * Kab); Paab = Paab + db * (b1 * Caab + b0 * Kaab); Pabb = Pabb + da * (a1 * Cabb + a0 * Kabb);
    }
    P1 = P1 / 2.0; Pa = Pa / 6.0; Paa = Paa / 12.0; Paaa = Paaa / 20.0;
    Pb = Pb / -6.0; Pbb = Pbb / -12.0; Pbbb = Pbbb / -20.0;
    Pab = Pab / 24.0; Paab = Paab / 60.0; Pabb = Pabb / -60.0;
    // comp_face_integrals (bodies.py:308-345)
    const T nC = pick3(C, 0, 1, nrm[0], nrm[1], nrm[2]), nA = pick3(A, 0, 1, nrm[0], nrm[1], nrm[2]), nB = pick3(B, 0, 1, nrm[0], nrm[1], nrm[2]);
    const T k1 = 1.0 / nC, k2 = k1 * k1, k3 = k2 * k1, k4 = k3 * k1;
    const T Fa = k1 * Pa, Fb = k1 * Pb, Fc = -k2 * (nA * Pa + nB * Pb + w * P1);
    const T Faa = k1 * Paa, Fbb = k1 * Pbb;
    const T Fcc = k3 * (nA * nA * Paa + 2.0 * nA * nB * Pab + nB * nB * Pbb + w * (2.0 * (nA * Pa + nB * Pb) + w * P1));
    const T Faaa = k1 * Paaa, Fbbb = k1 * Pbbb;
    const T Fccc = -k4 * (nA * nA * nA * Paaa + 3.0 * nA * nA * nB * Paab + 3.0 * nA * nB * nB * Pabb + nB * nB * nB * Pbbb
                          + 3.0 * w * (nA * nA * Paa + 2.0 * nA * nB * Pab + nB * nB * Pbb)
                          + w * w * (3.0 * (nA * Pa + nB * Pb) + w * P1));
    const T Faab = k1 * Paab, Fbbc = -k2 * (nA * Pabb + nB * Pbbb + w * Pbb);
    const T Fcca = k3 * (nA * nA * Paaa + 2.0 * nA * nB * Paab + nB * nB * Pabb + w * (2.0 * (nA * Paa + nB * Pab) + w * Pa));
    // comp_volume_integrals (bodies.py:348-377): the x-coordinate term of T0, per-axis scatter of the others
    c[0] = nrm[0] * pick3(0, A, B, Fa, Fb, Fc);
    for (int ax = 0; ax < 3; ++ax) {
        c[1 + ax] = pick3(ax, A, B, nA * Faa, nB * Fbb, nC * Fcc);
        c[4 + ax] = pick3(ax, A, B, nA * Faaa, nB * Fbbb, nC * Fccc);
        c[7 + ax] = pick3(ax, A, B, nA * Faab, nB * Fbbc, nC * Fcca);
    }
}

// workgroup-wide ordered sums of the ten integrals over the faces of one mesh (fixed tree: reproducible)
__device__ inline void mesh_totals(const double *V, const int *F, int nf, double *red, double *tot)
{
    const int tid = threadIdx.x;
    double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int f = tid; f < nf; f += 256) {
        double v[3][3], c[10];
        for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) v[k][i] = V[(size_t)F[3 * f + k] * 3 + i];
        face_integrals(v, c);
        for (int q = 0; q < 10; ++q) acc[q] += c[q];
    }
    for (int q = 0; q < 10; ++q) {
        red[tid] = acc[q];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
        tot[q] = red[0];
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) mesh_inertia_kernel(const double *verts, const int *faces, const int *voff,
                                                          const int *foff, const int *nfs, const double *mass, double *J,
                                                          double *vol)
{
    __shared__ double red[256];
    const int m = blockIdx.x, tid = threadIdx.x, nf = nfs[m];
    const double *V = verts + (size_t)voff[m] * 3;
    const int *F = faces + (size_t)foff[m] * 3;
    double tot[10];
    mesh_totals(V, F, nf, red, tot);
    if (tid == 0) {
        const double T0 = tot[0];
        const double T2[3] = {tot[4] / 3.0, tot[5] / 3.0, tot[6] / 3.0}, TP[3] = {tot[7] / 2.0, tot[8] / 2.0, tot[9] / 2.0};
        const double density = mass[m] / T0;
        double *o = J + (size_t)m * 9;
        o[0] = density * (T2[1] + T2[2]); o[4] = density * (T2[2] + T2[0]); o[8] = density * (T2[0] + T2[1]);
        o[1] = o[3] = -density * TP[0];
        o[5] = o[7] = -density * TP[1];
        o[6] = o[2] = -density * TP[2];
        if (vol) vol[m] = T0;
    }
}

// d (sum_ab gJ_ab J_ab) / d verts.  J = rho (sums of T2) / -rho TP with rho = mass / T0 (bodies.py:380-395): the adjoint
// of the ten totals is a handful of scalars; every face then differentiates its own contribution with dual numbers
// (three passes, one per vertex) and adds it to its vertices.  Atomic adds: the summation order over the faces that
// share a vertex is not fixed (world-construction gradient, last-bit differences between runs).
// Several workgroups per mesh (round 3; one workgroup took 3.5 ms for the 35 000 faces of a level-set body, a fifth of a
// config-5 iteration): each recomputes the ten totals for itself -- plain doubles, a thirteenth of the dual-number work of the
// faces it then differentiates -- so no scratch buffer and no second launch are needed.
__global__ void __launch_bounds__(256) mesh_inertia_bwd_kernel(const double *verts, const int *faces, int nf, double mass,
                                                              const double *gJ, double *gverts)
{
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double tot[10];
    mesh_totals(verts, faces, nf, red, tot);
    const double T0 = tot[0], rho = mass / T0;
    const double T2[3] = {tot[4] / 3.0, tot[5] / 3.0, tot[6] / 3.0}, TP[3] = {tot[7] / 2.0, tot[8] / 2.0, tot[9] / 2.0};
    const double Jv[9] = {rho * (T2[1] + T2[2]), -rho * TP[0], -rho * TP[2], -rho * TP[0], rho * (T2[2] + T2[0]), -rho * TP[1],
                          -rho * TP[2], -rho * TP[1], rho * (T2[0] + T2[1])};
    double L = 0.0;
    for (int e = 0; e < 9; ++e) L += gJ[e] * Jv[e];
    double gT[10];
    gT[0] = -L / T0;
    gT[1] = gT[2] = gT[3] = 0.0;
    gT[4] = rho / 3.0 * (gJ[4] + gJ[8]); gT[5] = rho / 3.0 * (gJ[0] + gJ[8]); gT[6] = rho / 3.0 * (gJ[0] + gJ[4]);
    gT[7] = -rho / 2.0 * (gJ[1] + gJ[3]); gT[8] = -rho / 2.0 * (gJ[5] + gJ[7]); gT[9] = -rho / 2.0 * (gJ[2] + gJ[6]);
    typedef Dual<3> D;
    for (int f = blockIdx.x * 256 + tid; f < nf; f += gridDim.x * 256) {
        for (int vs = 0; vs < 3; ++vs) {
            D v[3][3], c[10];
            for (int k = 0; k < 3; ++k)
                for (int i = 0; i < 3; ++i) { v[k][i] = D(verts[(size_t)faces[3 * f + k] * 3 + i]); if (k == vs) v[k][i].d[i] = 1.0; }
            face_integrals(v, c);
            for (int i = 0; i < 3; ++i) {
                double g = 0.0;
                for (int q = 0; q < 10; ++q) g += gT[q] * c[q].d[i];
                atomicAdd(&gverts[(size_t)faces[3 * f + vs] * 3 + i], g);
            }
        }
    }
}

// self-test of geom.h's shared-reciprocal triple division against three IEEE divisions (bit patterns compared)
__global__ void __launch_bounds__(256) div3_selftest_kernel(const double *num, const double *den, int n, int *mismatches)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double a[3] = {num[3 * (size_t)i], num[3 * (size_t)i + 1], num[3 * (size_t)i + 2]}, d = den[i];
    double q[3];
    div3(a, d, q);
    int bad = 0;
    for (int k = 0; k < 3; ++k) {
        volatile double ref = a[k] / d;       // volatile: keep it a plain division
        bad += __double_as_longlong(q[k]) != __double_as_longlong((double)ref);
    }
    if (bad) atomicAdd(mismatches, bad);
}

__global__ void __launch_bounds__(256) sqrt_selftest_kernel(const double *x, int n, int *mismatches)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    volatile double ref = sqrt(x[i]);
    if (__double_as_longlong(t_sqrt(x[i])) != __double_as_longlong((double)ref)) atomicAdd(mismatches, 1);
}

}  // namespace

extern "C" {

int dss_selftest_sqrt(const double *x, int n, int *mismatches, void *stream)
{
    if (!x || !mismatches || n <= 0) return DSS_E_BADARG;
    hipLaunchKernelGGL(sqrt_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, n, mismatches);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}

int dss_mesh_inertia_backward(const double *verts, const int *faces, int nv, int nf, double mass, const double *grad_J,
                              double *grad_verts, void *stream)
{
    if (!verts || !faces || !grad_J || !grad_verts || nv <= 0 || nf <= 0) return DSS_E_BADARG;
    (void)hipMemsetAsync(grad_verts, 0, (size_t)nv * 3 * sizeof(double), (hipStream_t)stream);
    int grid = (nf + 255) / 256;
    if (grid > 32) grid = 32;
    hipLaunchKernelGGL(mesh_inertia_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, verts, faces, nf, mass, grad_J, grad_verts);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}

int dss_selftest_div3(const double *num, const double *den, int n, int *mismatches, void *stream)
{
    if (!num || !den || !mismatches || n <= 0) return DSS_E_BADARG;
    hipLaunchKernelGGL(div3_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, num, den, n, mismatches);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}

int dss_sdf_query(int shape_type, const double *prm, const double *pts, int n, double *sdf, double *grad,
                  unsigned char *overlap_mask, void *stream)
{
    if (!prm || !pts || !sdf || n <= 0) return DSS_E_BADARG;
    if (shape_type < DSS_SHAPE_BOX || shape_type > DSS_SHAPE_BOWL) return DSS_E_UNSUPPORTED;
    hipLaunchKernelGGL(sdf_query_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, shape_type, prm[0], prm[1],
                       prm[2], prm[3], pts, n, sdf, grad, overlap_mask);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}

int dss_grid_sdf_query(const double *grid, int n0, int n1, int n2, double scale, const double *pts, int n, double *sdf,
                       double *grad, unsigned char *overlap_mask, void *stream)
{
    if (!grid || !pts || !sdf || n <= 0 || n0 < 2 || n1 < 2 || n2 < 2 || !(scale > 0.0)) return DSS_E_BADARG;
    hipLaunchKernelGGL(grid_sdf_query_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, grid, n0, n1, n2, scale, pts,
                       n, sdf, grad, overlap_mask);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}

int dss_mesh_inertia(const double *verts, const int *faces, const int *mesh_voff, const int *mesh_foff, const int *mesh_nf,
                     int nmesh, const double *mass, double *J, double *volume, void *stream)
{
    if (!verts || !faces || !mesh_voff || !mesh_foff || !mesh_nf || !mass || !J || nmesh <= 0) return DSS_E_BADARG;
    hipLaunchKernelGGL(mesh_inertia_kernel, dim3(nmesh), dim3(256), 0, (hipStream_t)stream, verts, faces, mesh_voff, mesh_foff,
                       mesh_nf, mass, J, volume);
    return hipGetLastError() == hipSuccess ? DSS_OK : DSS_E_UNSUPPORTED;
}

}  // extern "C"
