// geom.h -- scalar-generic rigid-body / SDF geometry for the stepper kernels.
//
// Every function is a template on the scalar type T: `double` in the forward kernels and
// Dual<N> (value + N tangents) in the backward kernels, which obtain the vector-Jacobian
// product of a small stage (contact geometry, pose integration, world-frame inertia) by
// seeding its inputs and contracting the output tangents with the incoming adjoint.
// The derivative conventions follow what torch.autograd does in the reference for the
// non-smooth pieces (abs'(0) = 0, clamp passes the gradient on the boundary, max/min route
// the gradient to one arg-max, F.normalize clamps the norm at 1e-12).
//
// Conventions: quaternions are real-first (w,x,y,z); pose = [q(4), x(3)];
// generalized velocity = [omega(3), v(3)]; pytorch3d==0.7.5 semantics (restated, not vendored):
//   so3_exponential_map clamps |v|^2 at eps = 1e-4 before the square root,
//   quaternion_multiply standardises to a non-negative real part,
//   matrix_to_quaternion picks the best conditioned of four candidates.
#pragma once
#include <math.h>

#include "dss_device.h"

// The reference breaks exact ties (arg-min vertex of a flat triangle, Laplacian comparison of two flat
// faces) on the last bit of un-fused IEEE arithmetic.  Translation units that include this header keep
// mul/add un-contracted so the device takes the same branches as the CPU reference wherever the inputs
// agree bit for bit (the LCP kernels do not include it and keep FMA contraction).
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace dss {

template <int N> struct Dual {
    double v;
    double d[N];
    __host__ __device__ Dual() : v(0.0) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
    __host__ __device__ Dual(double x) : v(x) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
};

__host__ __device__ inline double val(double x) { return x; }
template <int N> __host__ __device__ inline double val(const Dual<N> &x) { return x.v; }

#define DSS_DUAL_BIN(op, expr_v, expr_d)                                                                  \
    template <int N> __host__ __device__ inline Dual<N> operator op(const Dual<N> &a, const Dual<N> &b)   \
    {                                                                                                      \
        Dual<N> r;                                                                                         \
        r.v = expr_v;                                                                                      \
        for (int i = 0; i < N; ++i) r.d[i] = expr_d;                                                       \
        return r;                                                                                          \
    }
DSS_DUAL_BIN(+, a.v + b.v, a.d[i] + b.d[i])
DSS_DUAL_BIN(-, a.v - b.v, a.d[i] - b.d[i])
DSS_DUAL_BIN(*, a.v * b.v, a.d[i] * b.v + a.v * b.d[i])
DSS_DUAL_BIN(/, a.v / b.v, (a.d[i] - (a.v / b.v) * b.d[i]) / b.v)
#undef DSS_DUAL_BIN
template <int N> __host__ __device__ inline Dual<N> operator+(const Dual<N> &a, double b) { Dual<N> r = a; r.v += b; return r; }
template <int N> __host__ __device__ inline Dual<N> operator+(double b, const Dual<N> &a) { return a + b; }
template <int N> __host__ __device__ inline Dual<N> operator-(const Dual<N> &a, double b) { Dual<N> r = a; r.v -= b; return r; }
template <int N> __host__ __device__ inline Dual<N> operator-(double b, const Dual<N> &a)
{
    Dual<N> r; r.v = b - a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r;
}
template <int N> __host__ __device__ inline Dual<N> operator-(const Dual<N> &a)
{
    Dual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r;
}
template <int N> __host__ __device__ inline Dual<N> operator*(const Dual<N> &a, double b)
{
    Dual<N> r; r.v = a.v * b; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b; return r;
}
template <int N> __host__ __device__ inline Dual<N> operator*(double b, const Dual<N> &a) { return a * b; }
template <int N> __host__ __device__ inline Dual<N> operator/(const Dual<N> &a, double b) { return a * (1.0 / b); }
template <int N> __host__ __device__ inline Dual<N> operator/(double b, const Dual<N> &a) { return Dual<N>(b) / a; }

// IEEE square root.  The device's sqrt(double) is v_rsq_f64 + a Goldschmidt step + two residual corrections, wrapped in
// a range rescale for arguments below 2^-767; the squared lengths of this file are zero or far above that, so the same
// sequence without the rescale returns the same bits (checked against sqrt() on the device, dss_selftest_sqrt) and
// saves a third of the instructions of the commonest transcendental of an SDF query.
__host__ __device__ inline double t_sqrt(double x)
{
#if !defined(DSS_EMU) && defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(x);
    const double g0 = x * y, h0 = 0.5 * y;
    const double r0 = __builtin_fma(-h0, g0, 0.5);
    const double g1 = __builtin_fma(g0, r0, g0), h1 = __builtin_fma(h0, r0, h0);
    const double g2 = __builtin_fma(__builtin_fma(-g1, g1, x), h1, g1);
    const double g3 = __builtin_fma(__builtin_fma(-g2, g2, x), h1, g2);
    return (x == 0.0 || x == INFINITY) ? x : g3;
#else
    return sqrt(x);
#endif
}
template <int N> __host__ __device__ inline Dual<N> t_sqrt(const Dual<N> &a)
{
    Dual<N> r; r.v = sqrt(a.v);
    const double k = a.v > 0.0 ? 0.5 / r.v : 0.0;
    for (int i = 0; i < N; ++i) r.d[i] = k * a.d[i];
    return r;
}
__host__ __device__ inline double t_sin(double x) { return sin(x); }
__host__ __device__ inline double t_cos(double x) { return cos(x); }
template <int N> __host__ __device__ inline Dual<N> t_sin(const Dual<N> &a)
{
    Dual<N> r; r.v = sin(a.v); const double c = cos(a.v);
    for (int i = 0; i < N; ++i) r.d[i] = c * a.d[i];
    return r;
}
template <int N> __host__ __device__ inline Dual<N> t_cos(const Dual<N> &a)
{
    Dual<N> r; r.v = cos(a.v); const double s = -sin(a.v);
    for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i];
    return r;
}
// torch.abs: d/dx = sign(x), 0 at 0
__host__ __device__ inline double t_abs(double x) { return fabs(x); }
template <int N> __host__ __device__ inline Dual<N> t_abs(const Dual<N> &a)
{
    const double s = a.v > 0.0 ? 1.0 : (a.v < 0.0 ? -1.0 : 0.0);
    Dual<N> r; r.v = fabs(a.v);
    for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i];
    return r;
}
// clamp(x, min=lo): gradient passes when x >= lo (torch.clamp backward mask is x >= min)
template <class T> __host__ __device__ inline T t_clamp_min(const T &x, double lo) { return val(x) >= lo ? x : T(lo); }
template <class T> __host__ __device__ inline T t_clamp_max(const T &x, double hi) { return val(x) <= hi ? x : T(hi); }
// select by value; ties go to the first argument (index-0-wins like torch.max(dim).indices on CPU)
template <class T> __host__ __device__ inline T t_max(const T &a, const T &b) { return val(b) > val(a) ? b : a; }
template <class T> __host__ __device__ inline T t_min(const T &a, const T &b) { return val(b) < val(a) ? b : a; }
// elementwise torch.max(a, b) / torch.maximum: at an exact tie autograd gives each argument half the gradient
__host__ __device__ inline double t_maximum(double a, double b) { return a > b ? a : b; }
template <int N> __host__ __device__ inline Dual<N> t_maximum(const Dual<N> &a, const Dual<N> &b)
{
    if (a.v > b.v) return a;
    if (b.v > a.v) return b;
    return (a + b) * 0.5;
}

template <class T> struct V3 { T x[3]; };

template <class T> __host__ __device__ inline void cross(const T *a, const T *b, T *o)
{
    T o0 = a[1] * b[2] - a[2] * b[1], o1 = a[2] * b[0] - a[0] * b[2], o2 = a[0] * b[1] - a[1] * b[0];
    o[0] = o0; o[1] = o1; o[2] = o2;
}
template <class T> __host__ __device__ inline T dot(const T *a, const T *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <class T> __host__ __device__ inline T norm3(const T *a) { return t_sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
// Three IEEE quotients by one denominator.  On the device a double division is the sequence
//   r = rcp(d); two Newton steps on r; q = n r; e = fma(-d, q, n); q = fma(e, r, q)
// wrapped in range scaling / special-case fix-ups (v_div_scale, v_div_fmas, v_div_fixup).  For the operands of this
// file (lengths and coordinates between 1e-12 and 1e3, never 0/0, inf or nan) the scaling is the identity, so running
// the reciprocal refinement once and the three-instruction tail per numerator yields bit for bit the same quotients
// as three `/` -- 17 instead of 33 instructions per normalisation, the commonest operation of an SDF query.
template <class T> __host__ __device__ inline void div3(const T *num, const T &den, T *out)
{
    for (int i = 0; i < 3; ++i) out[i] = num[i] / den;
}
__host__ __device__ inline void div3(const double *num, const double &den, double *out)
{
#if !defined(DSS_EMU) && defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(den);
    r = __builtin_fma(r, __builtin_fma(-den, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-den, r, 1.0), r);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double n = num[i], q = n * r;
        out[i] = __builtin_fma(__builtin_fma(-den, q, n), r, q);   // out may alias num
    }
#else
    for (int i = 0; i < 3; ++i) out[i] = num[i] / den;
#endif
}
// torch.nn.functional.normalize(v, dim): v / max(||v||, 1e-12)
template <class T> __host__ __device__ inline void normalize(const T *a, T *o)
{
    T n = norm3(a);
    if (val(n) < 1e-12) n = T(1e-12);
    T q[3];
    div3(a, n, q);
    for (int i = 0; i < 3; ++i) o[i] = q[i];
}

// ---- quaternions (pytorch3d.transforms semantics) -------------------------------------------
template <class T> __host__ __device__ inline void quat_raw_mul(const T *a, const T *b, T *o)
{
    T ow = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    T ox = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    T oy = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    T oz = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    o[0] = ow; o[1] = ox; o[2] = oy; o[3] = oz;
}
template <class T> __host__ __device__ inline void quat_mul(const T *a, const T *b, T *o)
{
    quat_raw_mul(a, b, o);
    if (val(o[0]) < 0.0) for (int i = 0; i < 4; ++i) o[i] = -o[i];
}
template <class T> __host__ __device__ inline void quat_inv(const T *q, T *o)
{
    o[0] = q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = -q[3];
}
// quaternion_apply(q, p) = (q * (0,p) * q^-1)[1:]   (no normalisation, as in pytorch3d)
template <class T> __host__ __device__ inline void quat_apply(const T *q, const T *p, T *o)
{
    T pq[4] = {T(0.0), p[0], p[1], p[2]}, qi[4], t[4], r[4];
    quat_inv(q, qi);
    quat_raw_mul(q, pq, t);
    quat_raw_mul(t, qi, r);
    o[0] = r[1]; o[1] = r[2]; o[2] = r[3];
}
// The same for plain doubles with the zero real part of (0, p) folded away: a0 * 0 and "+ 0" drop out of quat_raw_mul's
// left-to-right sums without changing any other bit (0 - x = -x and x + 0 = x exactly; only the sign of an exact zero can
// differ, which nothing downstream looks at), and the real part of the second product is never formed.  12 multiplies and
// 7 additions fewer per call, in the most frequent operation of the narrow phase.
__host__ __device__ inline void quat_apply(const double *q, const double *p, double *o)
{
    const double tw = -q[1] * p[0] - q[2] * p[1] - q[3] * p[2];
    const double tx = q[0] * p[0] + q[2] * p[2] - q[3] * p[1];
    const double ty = q[0] * p[1] - q[1] * p[2] + q[3] * p[0];
    const double tz = q[0] * p[2] + q[1] * p[1] - q[2] * p[0];
    const double i0 = q[0], i1 = -q[1], i2 = -q[2], i3 = -q[3];      // q^-1
    const double ox = tw * i1 + tx * i0 + ty * i3 - tz * i2;
    const double oy = tw * i2 - tx * i3 + ty * i0 + tz * i1;
    const double oz = tw * i3 + tx * i2 - ty * i1 + tz * i0;
    o[0] = ox; o[1] = oy; o[2] = oz;
}
template <class T> __host__ __device__ inline void quat_apply_inv(const T *q, const T *p, T *o)
{
    T qi[4];
    quat_inv(q, qi);
    quat_apply(qi, p, o);
}
template <class T> __host__ __device__ inline void quat_to_mat(const T *q, T *R /*[9] row-major*/)
{
    const T r = q[0], i = q[1], j = q[2], k = q[3];
    const T two_s = 2.0 / (r * r + i * i + j * j + k * k);
    R[0] = 1.0 - two_s * (j * j + k * k); R[1] = two_s * (i * j - k * r); R[2] = two_s * (i * k + j * r);
    R[3] = two_s * (i * j + k * r); R[4] = 1.0 - two_s * (i * i + k * k); R[5] = two_s * (j * k - i * r);
    R[6] = two_s * (i * k - j * r); R[7] = two_s * (j * k + i * r); R[8] = 1.0 - two_s * (i * i + j * j);
}
// so3_exponential_map(v, eps=1e-4)
template <class T> __host__ __device__ inline void so3_exp(const T *w, T *R)
{
    T nr = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    nr = t_clamp_min(nr, 1e-4);
    const T ang = t_sqrt(nr), inv = 1.0 / ang;
    const T f1 = inv * t_sin(ang), f2 = inv * inv * (1.0 - t_cos(ang));
    const T K[9] = {T(0.0), -w[2], w[1], w[2], T(0.0), -w[0], -w[1], w[0], T(0.0)};
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            T k2 = K[3 * a] * K[b] + K[3 * a + 1] * K[3 + b] + K[3 * a + 2] * K[6 + b];
            R[3 * a + b] = f1 * K[3 * a + b] + f2 * k2 + (a == b ? 1.0 : 0.0);
        }
}
template <class T> __host__ __device__ inline T sqrt_pos(const T &x) { return val(x) > 0.0 ? t_sqrt(x) : T(0.0); }
template <class T> __host__ __device__ inline void mat_to_quat(const T *m, T *q)
{
    const T qa[4] = {sqrt_pos(1.0 + m[0] + m[4] + m[8]), sqrt_pos(1.0 + m[0] - m[4] - m[8]),
                     sqrt_pos(1.0 - m[0] + m[4] - m[8]), sqrt_pos(1.0 - m[0] - m[4] + m[8])};
    int best = 0;
    for (int i = 1; i < 4; ++i) if (val(qa[i]) > val(qa[best])) best = i;
    T c[4];
    switch (best) {
    case 0: c[0] = qa[0] * qa[0]; c[1] = m[7] - m[5]; c[2] = m[2] - m[6]; c[3] = m[3] - m[1]; break;
    case 1: c[0] = m[7] - m[5]; c[1] = qa[1] * qa[1]; c[2] = m[3] + m[1]; c[3] = m[2] + m[6]; break;
    case 2: c[0] = m[2] - m[6]; c[1] = m[3] + m[1]; c[2] = qa[2] * qa[2]; c[3] = m[5] + m[7]; break;
    default: c[0] = m[3] - m[1]; c[1] = m[6] + m[2]; c[2] = m[7] + m[5]; c[3] = qa[3] * qa[3]; break;
    }
    T den = qa[best];
    if (val(den) < 0.1) den = T(0.1);
    den = den * 2.0;
    for (int i = 0; i < 4; ++i) q[i] = c[i] / den;
    if (val(q[0]) < 0.0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
}

// Body3D.move (physics3d/bodies.py:488-496): q <- quat(exp(w dt)) (x) q ; x <- x + v dt
template <class T> __host__ __device__ inline void integrate_pose(const T *pose, const T *vel, const T &dt, T *out)
{
    T w[3] = {vel[0] * dt, vel[1] * dt, vel[2] * dt}, R[9], dq[4];
    so3_exp(w, R);
    mat_to_quat(R, dq);
    quat_mul(dq, pose, out);
    for (int i = 0; i < 3; ++i) out[4 + i] = pose[4 + i] + vel[3 + i] * dt;
}

// world-frame rotational inertia R I R^T (physics3d/bodies.py:509-511)
template <class T> __host__ __device__ inline void world_inertia(const T *q, const T *Ib /*[9]*/, T *out /*[9]*/)
{
    T R[9], t[9];
    quat_to_mat(q, R);
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) t[3 * a + b] = R[3 * a] * Ib[b] + R[3 * a + 1] * Ib[3 + b] + R[3 * a + 2] * Ib[6 + b];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) out[3 * a + b] = t[3 * a] * R[3 * b] + t[3 * a + 1] * R[3 * b + 1] + t[3 * a + 2] * R[3 * b + 2];
}

// ---- primitive SDFs (physics3d/bodies.py:38-185) and SDF3D.query_sdfs (:721-760) ------------
// The stepper's two hot kernels (narrow phase, contact adjoint) are compiled twice: a lean variant that knows only box /
// sphere / cylinder (DSS_ALL_SHAPES 0: the remaining primitives cost it registers it does not have) and a variant with
// every primitive, picked at launch by DssWorld.shape_rare.  The shape code sits in an inline namespace per variant, so
// the two sets of inline functions are distinct entities.
#ifndef DSS_ALL_SHAPES
#define DSS_ALL_SHAPES 1
#endif
#if DSS_ALL_SHAPES
inline namespace shapes_all {
#else
inline namespace shapes_lean {
#endif
enum ShapeType { SHAPE_BOX = 0, SHAPE_SPHERE = 1, SHAPE_CYLINDER = 2, SHAPE_BOX_ROUNDED = 3, SHAPE_BRICK = 4, SHAPE_BOWL = 5, SHAPE_IGR = 6, SHAPE_GRID = 7 };

template <class T> struct Shape {
    int type;
    T prm[3];   // box / rounded box / brick: dims ; sphere: rad ; cylinder: rad, height (axis = body z) ; bowl: r, d
    T scale;    // box family: 1.5*max(dims)/2 ; sphere: 1.5*rad ; cylinder: 1.5*max(rad, height/2) ; bowl: 1.3333*(r+d)
    T hd[3];    // unit-frame constants hoisted out of the queries (invariant per body), see make_shape
#if DSS_ALL_SHAPES
    T hr;       // rounded box / brick: corner radius / scale
    // neural SDF (SHAPE_IGR) in the reverse sweep: records of the network's value and derivatives at the points this body is
    // queried at for ONE contact (igr_lin below), evaluated beforehand on the matrix cores; NULL elsewhere
    const double *lin;
    // voxel-grid SDF (SHAPE_GRID, SDFGrid3D bodies.py:203-257, 763-775): samples over the unit cube, x slowest
    const double *grid;
    int gn[3];
#endif
};
// aux: the corner radius r of SDFBoxRounded / SDFBrick (a constant of the body; bodies.py:857-885), unused otherwise
template <class T> __host__ __device__ inline void make_shape(Shape<T> &s, int type, const T *prm, double aux = 0.0)
{
    s.type = type;
    for (int i = 0; i < 3; ++i) s.prm[i] = prm[i];
#if DSS_ALL_SHAPES
    s.hr = T(0.0);
    s.lin = nullptr;
    s.grid = nullptr;
    s.gn[0] = s.gn[1] = s.gn[2] = 0;
#endif
    if (type == SHAPE_BOX) {
        s.scale = t_max(t_max(prm[0], prm[1]), prm[2]) * 1.5 / 2.0;
        for (int i = 0; i < 3; ++i) s.hd[i] = (prm[i] / s.scale) / 2.0;
#if DSS_ALL_SHAPES
    } else if (type == SHAPE_BOX_ROUNDED) {   // bodies.py:857-870: params r/scale, (dims - 2 r)/scale -> box_sdf halves them
        s.scale = t_max(t_max(prm[0], prm[1]), prm[2]) * 1.5 / 2.0;
        for (int i = 0; i < 3; ++i) s.hd[i] = ((prm[i] - 2.0 * aux) / s.scale) / 2.0;
        s.hr = T(aux) / s.scale;
    } else if (type == SHAPE_BRICK) {         // bodies.py:873-885, brick_sdf :161-163: half_dims = dims/2 ; half_dims[:2] -= r
        s.scale = t_max(t_max(prm[0], prm[1]), prm[2]) * 1.5 / 2.0;
        s.hr = T(aux) / s.scale;
        for (int i = 0; i < 3; ++i) s.hd[i] = (prm[i] / s.scale) / 2.0;
        s.hd[0] = s.hd[0] - s.hr; s.hd[1] = s.hd[1] - s.hr;
#endif
    } else if (type == SHAPE_CYLINDER) {   // bodies.py:913-921: scale = 1.5 max(rad, height/2), params rad/scale, height/scale
        s.scale = t_max(prm[0], prm[1] / 2.0) * 1.5;
        s.hd[0] = prm[0] / s.scale; s.hd[1] = (prm[1] / s.scale) / 2.0; s.hd[2] = T(0.0);
#if DSS_ALL_SHAPES
    } else if (type == SHAPE_BOWL) {       // bodies.py:1013-1027: scale = (r + d) * 1.3333, params r/scale, d/scale
        s.scale = (prm[0] + prm[1]) * 1.3333;
        s.hd[0] = prm[0] / s.scale; s.hd[1] = prm[1] / s.scale; s.hd[2] = T(0.0);
    } else if (type == SHAPE_IGR || type == SHAPE_GRID) {   // SDF3D with a network / a voxel grid (bodies.py:627-651, 763-775): aux = the given scale
        s.scale = T(aux);
        for (int i = 0; i < 3; ++i) s.hd[i] = T(0.0);
#endif
    } else {
        s.scale = prm[0] * 1.5;
        s.hd[0] = prm[0] / s.scale; s.hd[1] = T(0.0); s.hd[2] = T(0.0);
    }
}

#if DSS_ALL_SHAPES
// the same in the reference's unit frame: pu = the body's parameters / scale, au = r / scale (what the reference hands to
// sdf_func as *params, up to its own grouping: rounded box (r/scale, (dims - 2 r)/scale))
template <class T> __host__ __device__ inline void make_unit_shape(Shape<T> &s, int type, const T *pu, const T &au)
{
    s.type = type;
    for (int i = 0; i < 3; ++i) { s.prm[i] = pu[i]; s.hd[i] = T(0.0); }
    s.scale = T(1.0);
    s.hr = au;
    s.lin = nullptr;
    s.grid = nullptr;
    s.gn[0] = s.gn[1] = s.gn[2] = 0;
    if (type == SHAPE_BOX) for (int i = 0; i < 3; ++i) s.hd[i] = pu[i] / 2.0;          // box_sdf: half_dims = dims / 2
    else if (type == SHAPE_BOX_ROUNDED) for (int i = 0; i < 3; ++i) s.hd[i] = (pu[i] - au * 2.0) / 2.0;
    else if (type == SHAPE_BRICK) { for (int i = 0; i < 3; ++i) s.hd[i] = pu[i] / 2.0; s.hd[0] = s.hd[0] - au; s.hd[1] = s.hd[1] - au; }
    else if (type == SHAPE_CYLINDER) { s.hd[0] = pu[0]; s.hd[1] = pu[1] / 2.0; }
    else if (type == SHAPE_BOWL) { s.hd[0] = pu[0]; s.hd[1] = pu[1]; }
    else s.hd[0] = pu[0];
}

// box_sdf_grad (bodies.py:51-72) for half extents hd: failsafe diagonal normals on ties; normalised twice
// (once by the function, once more by query_sdfs, bodies.py:748)
template <class T> __host__ __device__ __forceinline__ void box_unit_grad(const T *p, const T *hd, T *g)
{
    T q[3], mg[3], nm[3], go[3];
    for (int i = 0; i < 3; ++i) q[i] = t_abs(p[i]) - hd[i];
    const T md = t_max(t_max(q[0], q[1]), q[2]);
    for (int i = 0; i < 3; ++i) mg[i] = t_maximum(q[i], T(0.0));   // torch.max(q, zeros): ties split
    normalize(mg, nm);
    for (int i = 0; i < 3; ++i) {
        const double sg = val(p[i]) < 0.0 ? -1.0 : 1.0;
        const double md_dir = (val(md) <= 0.0 && val(q[i]) == val(md)) ? 1.0 : 0.0;
        go[i] = (nm[i] + md_dir) * sg;
    }
    T g1[3];
    normalize(go, g1);
    normalize(g1, g);
}

#endif

// value and (normalised) gradient of the unit-cube SDF at p = pts/scale, parameters prm/scale
template <class T> __host__ __device__ inline void sdf_unit(const Shape<T> &s, const T *p, T &phi, T *g, bool want_grad)
{
#if DSS_ALL_SHAPES
    if (s.type == SHAPE_BOX || s.type == SHAPE_BOX_ROUNDED) {
#else
    if (s.type == SHAPE_BOX) {
#endif
        T q[3], m[3];
        for (int i = 0; i < 3; ++i) q[i] = t_abs(p[i]) - s.hd[i];
        const T md = t_max(t_max(q[0], q[1]), q[2]);
        for (int i = 0; i < 3; ++i) m[i] = t_clamp_min(q[i], 0.0);
        phi = norm3(m) + t_clamp_max(md, 0.0);
#if DSS_ALL_SHAPES
        if (s.type == SHAPE_BOX_ROUNDED) phi = phi - s.hr;   // rounded_sdf (bodies.py:139-145); gradient of the base box
#endif
        if (want_grad) {
            // box_sdf_grad: failsafe diagonal normals on ties (bodies.py:51-72); m = max(q, 0)
            T mg[3], nm[3], go[3];
            for (int i = 0; i < 3; ++i) mg[i] = t_maximum(q[i], T(0.0));   // torch.max(q, zeros): ties split
            normalize(mg, nm);
            for (int i = 0; i < 3; ++i) {
                const double sg = val(p[i]) < 0.0 ? -1.0 : 1.0;
                const double md_dir = (val(md) <= 0.0 && val(q[i]) == val(md)) ? 1.0 : 0.0;
                go[i] = (nm[i] + md_dir) * sg;
            }
            T g1[3];
            normalize(go, g1);
            normalize(g1, g);  // query_sdfs normalises again (bodies.py:748)
        }
    } else if (s.type == SHAPE_CYLINDER) {
        // cylinder_sdf / cylinder_sdf_grad (bodies.py:85-124): a 2-D box SDF in (radial, axial) coordinates
        const T rho = t_sqrt(p[0] * p[0] + p[1] * p[1]);
        T q[2] = {t_abs(rho) - s.hd[0], t_abs(p[2]) - s.hd[1]};
        const T md = t_max(q[0], q[1]);
        T m[2] = {t_clamp_min(q[0], 0.0), t_clamp_min(q[1], 0.0)};
        phi = t_sqrt(m[0] * m[0] + m[1] * m[1]) + t_clamp_max(md, 0.0);
        if (want_grad) {
            T nm = t_sqrt(m[0] * m[0] + m[1] * m[1]);
            if (val(nm) < 1e-12) nm = T(1e-12);
            const double inside = val(md) <= 0.0 ? 1.0 : 0.0;
            const T g0 = m[0] / nm + inside * (val(q[0]) == val(md) ? 1.0 : 0.0);
            const T g1 = m[1] / nm + inside * (val(q[1]) == val(md) ? 1.0 : 0.0);
            const double sg = val(p[2]) < 0.0 ? -1.0 : 1.0;
            T rxy = rho;                              // normalize(pts[:, :2])
            if (val(rxy) < 1e-12) rxy = T(1e-12);
            T go[3] = {g0 * (p[0] / rxy), g0 * (p[1] / rxy), g1 * sg}, g1n[3];
            normalize(go, g1n);
            normalize(g1n, g);  // query_sdfs normalises again (bodies.py:748)
        }
#if DSS_ALL_SHAPES
    } else if (s.type == SHAPE_BRICK) {
        // brick_sdf (bodies.py:160-177): rounded rectangle in (x, y), then a 2-D box of (that distance, |z| - hz)
        const T q0 = t_abs(p[0]) - s.hd[0], q1 = t_abs(p[1]) - s.hd[1], q2 = t_abs(p[2]) - s.hd[2];
        const T md01 = t_max(q0, q1);
        const T m0 = t_clamp_min(q0, 0.0), m1 = t_clamp_min(q1, 0.0);
        const T s01 = t_sqrt(m0 * m0 + m1 * m1) + t_clamp_max(md01, 0.0) - s.hr;
        const T md = t_max(s01, q2);
        const T n0 = t_clamp_min(s01, 0.0), n1 = t_clamp_min(q2, 0.0);
        phi = t_sqrt(n0 * n0 + n1 * n1) + t_clamp_max(md, 0.0);
        if (want_grad) {
            // SDFBrick passes grad_func = rounded_sdf_grad(box_sdf_grad) with params (dims/scale, r/scale): the wrapper
            // drops its FIRST parameter, so the reference's normal is box_sdf_grad(pts, r/scale), a cube of side
            // r/scale (bodies.py:148-154, 881-883).  Restated as is.
            const T h = s.hr / 2.0, hh[3] = {h, h, h};
            box_unit_grad(p, hh, g);
        }
    } else if (s.type == SHAPE_BOWL) {
        // bowl_sdf / bowl_sdf_grad (bodies.py:98-136).  Both shift pts[:, 2] by r/2 IN PLACE and query_sdfs hands them the
        // same tensor (bodies.py:746-748), so the gradient is evaluated at a point shifted twice.  Restated as is.
        const T r = s.hd[0], d = s.hd[1];
        const T z = p[2] - r / 2.0;
        const T rho = t_sqrt(p[0] * p[0] + p[1] * p[1]);
        {
            const T pn = t_sqrt(rho * rho + z * z);
            T a = val(z) < 0.0 ? pn : rho;
            a = t_abs(a - r) - d;
            const T m0 = t_maximum(a, T(0.0)), m1 = t_maximum(z, T(0.0));
            phi = t_sqrt(m0 * m0 + m1 * m1) + t_min(T(0.0), t_max(a, z));
        }
        if (want_grad) {
            const T z2 = z - r / 2.0;
            const T pn = t_sqrt(rho * rho + z2 * z2);
            T a = val(z2) < 0.0 ? pn : rho;
            a = t_abs(a - r) - d;
            const double dv = val(pn) - val(r), sg = dv > 0.0 ? 1.0 : (dv < 0.0 ? -1.0 : 0.0);
            T go[3] = {p[0] * sg, p[1] * sg, z2 * sg};
            if (val(z2) >= 0.0) {
                if (val(a) < 0.0) { go[0] = T(0.0); go[1] = T(0.0); }
                go[2] = t_abs(go[2]);
            }
            T g1[3];
            normalize(go, g1);
            normalize(g1, g);
        }
#endif
#if DSS_ALL_SHAPES
    } else if (s.type == SHAPE_GRID) {
        // grid_sdf / grid_sdf_grad (bodies.py:203-241): value = trilinear interpolation of the samples at index position
        // (p + 1)/2 (n - 1); gradient = the interpolated central-difference field (zero on the boundary layers), normalised
        // (twice, like every analytic gradient: once by grid_sdf_grad, once by query_sdfs).  `grid_interp` of the un-vendored
        // ev_sdf_utils is restated as plain trilinear interpolation.  To autograd the value's derivative w.r.t. the point IS
        // that normalised gradient (DiffGridSDF.backward, bodies.py:244-257) and the gradient itself carries no graph.
        const int n0 = s.gn[0], n1 = s.gn[1], n2 = s.gn[2];
        const int nn[3] = {n0, n1, n2};
        int i0[3];
        double w[3];
        for (int d = 0; d < 3; ++d) {
            const double ind = (val(p[d]) + 1.0) * 0.5 * (double)(nn[d] - 1);
            int c = (int)floor(ind);
            c = c < 0 ? 0 : (c > nn[d] - 2 ? nn[d] - 2 : c);
            i0[d] = c; w[d] = ind - (double)c;
        }
        const double *Gd = s.grid;
        auto at = [&](int i, int j, int k) { return Gd[((size_t)i * n1 + j) * n2 + k]; };
        double pv = 0.0, gv[3] = {0.0, 0.0, 0.0};
        for (int dx = 0; dx < 2; ++dx)
            for (int dy = 0; dy < 2; ++dy)
                for (int dz = 0; dz < 2; ++dz) {
                    const double wt = (dx ? w[0] : 1.0 - w[0]) * (dy ? w[1] : 1.0 - w[1]) * (dz ? w[2] : 1.0 - w[2]);
                    const int i = i0[0] + dx, j = i0[1] + dy, k = i0[2] + dz;
                    pv = pv + at(i, j, k) * wt;
                    const double cx = (i == 0 || i == n0 - 1) ? 0.0 : (at(i + 1, j, k) - at(i - 1, j, k)) / 2.0;
                    const double cy = (j == 0 || j == n1 - 1) ? 0.0 : (at(i, j + 1, k) - at(i, j - 1, k)) / 2.0;
                    const double cz = (k == 0 || k == n2 - 1) ? 0.0 : (at(i, j, k + 1) - at(i, j, k - 1)) / 2.0;
                    gv[0] = gv[0] + cx * wt; gv[1] = gv[1] + cy * wt; gv[2] = gv[2] + cz * wt;
                }
        double g1[3], g2[3];
        normalize(gv, g1);
        normalize(g1, g2);
        T acc = T(pv);
        for (int i = 0; i < 3; ++i) acc = acc + (p[i] - val(p[i])) * g1[i];
        phi = acc;
        if (want_grad) for (int i = 0; i < 3; ++i) g[i] = T(g2[i]);
    } else if (s.type == SHAPE_IGR) {
        // a network is never evaluated lane by lane: its queries go through the matrix-core rounds (narrowphase_igr.hip,
        // igr_mlp.hip).  Reaching this branch is a bug; make it loud.
        phi = T(NAN);
        if (want_grad) for (int i = 0; i < 3; ++i) g[i] = T(NAN);
#endif
    } else {
        const T n = norm3(p);
        phi = n - s.hd[0];
        if (want_grad) { T g1[3]; normalize(p, g1); normalize(g1, g); }
    }
}

#if DSS_ALL_SHAPES
// A neural SDF body's query as the reference's autograd sees it (SDF3D.query_sdfs, bodies.py:727-745: the input gradient is
// taken WITHOUT create_graph, so the normal is a constant; the value keeps the graph to the point and to the latent code):
//   phi(pt, latent) = phi0 + raw . (pt - pt0) + dlat . (latent - latent0),   g = nrm   (constant)
// record `rec` of s.lin: phi0, raw[3] = d phi / d pt, dlat[2] = d phi / d latent (both in world units), nrm[3].
constexpr int IGR_LIN = 9;
template <class T> __host__ __device__ inline void igr_lin(const Shape<T> &s, int rec, const T *pt, T &phi, T *g)
{
    const double *r = s.lin + IGR_LIN * rec;
    T acc = T(r[0]);
    for (int i = 0; i < 3; ++i) acc = acc + (pt[i] - val(pt[i])) * r[1 + i];
    for (int j = 0; j < 2; ++j) acc = acc + (s.prm[j] - val(s.prm[j])) * r[4 + j];
    phi = acc;
    for (int i = 0; i < 3; ++i) g[i] = T(r[6 + i]);
}
#endif

// SDF3D.query_sdfs: outside the [-scale, scale]^3 box: phi = scale, grad = 0
template <class T> __host__ __device__ inline bool query_sdf(const Shape<T> &s, const T *pt, T &phi, T *g, bool want_grad)
{
    const double sc = val(s.scale);
    const bool inside = fabs(val(pt[0])) <= sc && fabs(val(pt[1])) <= sc && fabs(val(pt[2])) <= sc;
    if (!inside) {
        phi = s.scale;  // sdfs = ones * scale
        if (want_grad) for (int i = 0; i < 3; ++i) g[i] = T(0.0);
        return false;
    }
    T p[3];
    div3(pt, s.scale, p);
    T u;
    sdf_unit(s, p, u, g, want_grad);
    phi = u * s.scale;
    return true;
}

}  // inline namespace shapes_*
}  // namespace dss
