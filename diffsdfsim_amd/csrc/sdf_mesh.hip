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

__global__ void __launch_bounds__(256) sdf_query_kernel(int type, double p0, double p1, double p2, const double *pts, int n,
                                                       double *sdf, double *grad, unsigned char *mask)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Shape<double> s;
    const double prm[3] = {p0, p1, p2};
    make_shape(s, type, prm);
    const double pt[3] = {pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]};
    double phi, g[3] = {0.0, 0.0, 0.0};
    const bool in = query_sdf(s, pt, phi, g, grad != nullptr);
    sdf[i] = phi;
    if (grad) { grad[3 * (size_t)i] = g[0]; grad[3 * (size_t)i + 1] = g[1]; grad[3 * (size_t)i + 2] = g[2]; }
    if (mask) mask[i] = in ? 1 : 0;
}

// projection integrals of one face over the (A, B) plane, comp_projection_integrals (bodies.py:260-305)
struct Proj { double P1, Pa, Paa, Paaa, Pb, Pbb, Pbbb, Pab, Paab, Pabb; };
__device__ inline void projection_integrals(const double v[3][3], int A, int B, Proj &o)
{
    double P1 = 0, Pa = 0, Paa = 0, Paaa = 0, Pb = 0, Pbb = 0, Pbbb = 0, Pab = 0, Paab = 0, Pabb = 0;
    for (int e = 0; e < 3; ++e) {
        const double a0 = v[e][A], b0 = v[e][B], a1 = v[(e + 1) % 3][A], b1 = v[(e + 1) % 3][B];
        const double da = a1 - a0, db = b1 - b0;
        const double a0_2 = a0 * a0, a0_3 = a0_2 * a0, a0_4 = a0_3 * a0, b0_2 = b0 * b0, b0_3 = b0_2 * b0, b0_4 = b0_3 * b0;
        const double a1_2 = a1 * a1, a1_3 = a1_2 * a1, b1_2 = b1 * b1, b1_3 = b1_2 * b1;
        const double C1 = a1 + a0, Ca = a1 * C1 + a0_2, Caa = a1 * Ca + a0_3, Caaa = a1 * Caa + a0_4;
        const double Cb = b1 * (b1 + b0) + b0_2, Cbb = b1 * Cb + b0_3, Cbbb = b1 * Cbb + b0_4;
        const double Cab = 3 * a1_2 + 2 * a1 * a0 + a0_2, Kab = a1_2 + 2 * a1 * a0 + 3 * a0_2;
        const double Caab = a0 * Cab + 4 * a1_3, Kaab = a1 * Kab + 4 * a0_3;
        const double Cabb = 4 * b1_3 + 3 * b1_2 * b0 + 2 * b1 * b0_2 + b0_3, Kabb = b1_3 + 2 * b1_2 * b0 + 3 * b1 * b0_2 + 4 * b0_3;
        P1 += db * C1; Pa += db * Ca; Paa += db * Caa; Paaa += db * Caaa;
        Pb += da * Cb; Pbb += da * Cbb; Pbbb += da * Cbbb;
        Pab += db * (b1 * Cab + b0 * Kab); Paab += db * (b1 * Caab + b0 * Kaab); Pabb += da * (a1 * Cabb + a0 * Kabb);
    }
    o.P1 = P1 / 2.0; o.Pa = Pa / 6.0; o.Paa = Paa / 12.0; o.Paaa = Paaa / 20.0;
    o.Pb = Pb / -6.0; o.Pbb = Pbb / -12.0; o.Pbbb = Pbbb / -20.0;
    o.Pab = Pab / 24.0; o.Paab = Paab / 60.0; o.Pabb = Pabb / -60.0;
}

__global__ void __launch_bounds__(256) mesh_inertia_kernel(const double *verts, const int *faces, const int *voff,
                                                          const int *foff, const int *nfs, const double *mass, double *J,
                                                          double *vol)
{
    __shared__ double red[256];
    const int m = blockIdx.x, tid = threadIdx.x, nf = nfs[m];
    const double *V = verts + (size_t)voff[m] * 3;
    const int *F = faces + (size_t)foff[m] * 3;
    double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // T0, T1[3], T2[3], TP[3] before the final /2, /3, /2
    for (int f = tid; f < nf; f += 256) {
        double v[3][3], nrm[3];
        for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) v[k][i] = V[(size_t)F[3 * f + k] * 3 + i];
        const double e1[3] = {v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2]};
        const double e2[3] = {v[2][0] - v[1][0], v[2][1] - v[1][1], v[2][2] - v[1][2]};
        cross(e1, e2, nrm);
        const double ln = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
        for (int i = 0; i < 3; ++i) nrm[i] /= ln;
        const double w = -(nrm[0] * v[0][0] + nrm[1] * v[0][1] + nrm[2] * v[0][2]);
        int C = 0;   // torch.argmax: first index of the maximum
        for (int i = 1; i < 3; ++i) if (fabs(nrm[i]) > fabs(nrm[C])) C = i;
        const int A = (C + 1) % 3, B = (A + 1) % 3;
        Proj P;
        projection_integrals(v, A, B, P);
        // comp_face_integrals (bodies.py:308-345)
        const double k1 = 1.0 / nrm[C], k2 = k1 * k1, k3 = k2 * k1, k4 = k3 * k1, nA = nrm[A], nB = nrm[B];
        const double Fa = k1 * P.Pa, Fb = k1 * P.Pb, Fc = -k2 * (nA * P.Pa + nB * P.Pb + w * P.P1);
        const double Faa = k1 * P.Paa, Fbb = k1 * P.Pbb;
        const double Fcc = k3 * (nA * nA * P.Paa + 2 * nA * nB * P.Pab + nB * nB * P.Pbb + w * (2 * (nA * P.Pa + nB * P.Pb) + w * P.P1));
        const double Faaa = k1 * P.Paaa, Fbbb = k1 * P.Pbbb;
        const double Fccc = -k4 * (nA * nA * nA * P.Paaa + 3 * nA * nA * nB * P.Paab + 3 * nA * nB * nB * P.Pabb + nB * nB * nB * P.Pbbb
                                   + 3 * w * (nA * nA * P.Paa + 2 * nA * nB * P.Pab + nB * nB * P.Pbb)
                                   + w * w * (3 * (nA * P.Pa + nB * P.Pb) + w * P.P1));
        const double Faab = k1 * P.Paab, Fbbc = -k2 * (nA * P.Pabb + nB * P.Pbbb + w * P.Pbb);
        const double Fcca = k3 * (nA * nA * P.Paaa + 2 * nA * nB * P.Paab + nB * nB * P.Pabb + w * (2 * (nA * P.Paa + nB * P.Pab) + w * P.Pa));
        // comp_volume_integrals (bodies.py:348-377): the x-coordinate term of T0, per-axis scatter of the others
        acc[0] += nrm[0] * (A == 0 ? Fa : (B == 0 ? Fb : Fc));
        acc[1 + A] += nA * Faa;  acc[1 + B] += nB * Fbb;  acc[1 + C] += nrm[C] * Fcc;
        acc[4 + A] += nA * Faaa; acc[4 + B] += nB * Fbbb; acc[4 + C] += nrm[C] * Fccc;
        acc[7 + A] += nA * Faab; acc[7 + B] += nB * Fbbc; acc[7 + C] += nrm[C] * Fcca;
    }
    double tot[10];
    for (int q = 0; q < 10; ++q) {
        red[tid] = acc[q];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
        tot[q] = red[0];
        __syncthreads();
    }
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
    if (shape_type != DSS_SHAPE_BOX && shape_type != DSS_SHAPE_SPHERE && shape_type != DSS_SHAPE_CYLINDER) return DSS_E_UNSUPPORTED;
    hipLaunchKernelGGL(sdf_query_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, shape_type, prm[0], prm[1],
                       prm[2], pts, n, sdf, grad, overlap_mask);
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
