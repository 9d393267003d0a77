// kkt_reg.h -- register-resident LU of the reduced KKT matrix (n <= 64): lane i owns row i.
//
// The LDS version (factor_K / solve_K in lcp_contact.hip) spends its time in latency chains: a shuffle
// reduction for the pivot, three barriers and dependent LDS read-modify-writes per elimination step.
// Here the matrix lives in VGPRs with compile-time indices (N is a template parameter, every loop is
// fully unrolled), the pivot row is broadcast with v_readlane through SGPRs and rows never move:
// partial pivoting is implicit (`step` = elimination step at which a row served as pivot).
//   row i, before it is a pivot:   a[k] (k < current step) = multiplier l_ik
//   row i, once pivot at step s:   a[j] (j > s) = U_sj,  a[s] = 1 / U_ss
// Pivot choice: largest |a_ik| among unused rows, compared in float precision (ties -> lowest lane).
#pragma once
#include "wave_utils.h"

namespace dss {

__device__ __forceinline__ double wave_bcast(double x, int src_uniform)
{
#if defined(DSS_EMU)
    return __shfl(x, src_uniform, WAVE);
#else
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src_uniform);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src_uniform);
    return __hiloint2double(hi, lo);
#endif
}
__device__ __forceinline__ int wave_uniform(int v)
{
#if defined(DSS_EMU)
    return v;
#else
    return __builtin_amdgcn_readfirstlane(v);
#endif
}
// max over the 64 lanes of a 32-bit key, result uniform.  gfx950: DPP row shifts + row broadcasts (the
// classic GCN reduction ladder: shr 1,2,3 | shr 4 | shr 8 | bcast15 | bcast31, total in lane 63) instead of
// six LDS-crossbar shuffles.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
#if defined(DSS_EMU)
    for (int o = WAVE / 2; o > 0; o >>= 1) { const unsigned w = __shfl_xor(v, o, WAVE); v = w > v ? w : v; }
    return v;
#else
#define DSS_DPP_MAX(ctrl, rmask, bmask)                                                                       \
    { const unsigned t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, bmask, false);   \
      v = t > v ? t : v; }
    DSS_DPP_MAX(0x111, 0xf, 0xf)   // row_shr:1
    DSS_DPP_MAX(0x112, 0xf, 0xf)   // row_shr:2
    DSS_DPP_MAX(0x113, 0xf, 0xf)   // row_shr:3
    DSS_DPP_MAX(0x114, 0xf, 0xe)   // row_shr:4
    DSS_DPP_MAX(0x118, 0xf, 0xc)   // row_shr:8
    DSS_DPP_MAX(0x142, 0xa, 0xf)   // row_bcast:15
    DSS_DPP_MAX(0x143, 0xc, 0xf)   // row_bcast:31
#undef DSS_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
#endif
}
// min / max of a double over the wavefront with the same DPP ladder (exact, order-independent); ~6x shorter
// dependent chain than six ds_bpermute shuffles, which matters with one wavefront per SIMD
#if defined(DSS_EMU)
__device__ __forceinline__ double wave_min_dpp(double v) { return wave_min(v); }
__device__ __forceinline__ double wave_max_dpp(double v) { return wave_max(v); }
#else
#define DSS_DPP_D(OP, ctrl, rmask, bmask)                                                                          \
    { const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), ctrl, rmask, bmask, false); \
      const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), ctrl, rmask, bmask, false); \
      v = OP(v, __hiloint2double(hi, lo)); }
#define DSS_DPP_LADDER(OP)                                                                                         \
    DSS_DPP_D(OP, 0x111, 0xf, 0xf) DSS_DPP_D(OP, 0x112, 0xf, 0xf) DSS_DPP_D(OP, 0x113, 0xf, 0xf)                   \
    DSS_DPP_D(OP, 0x114, 0xf, 0xe) DSS_DPP_D(OP, 0x118, 0xf, 0xc) DSS_DPP_D(OP, 0x142, 0xa, 0xf)                   \
    DSS_DPP_D(OP, 0x143, 0xc, 0xf)                                                                                 \
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
__device__ __forceinline__ double wave_min_dpp(double v) { DSS_DPP_LADDER(fmin) }
__device__ __forceinline__ double wave_max_dpp(double v) { DSS_DPP_LADDER(fmax) }
#undef DSS_DPP_LADDER
#undef DSS_DPP_D
#endif
__device__ __forceinline__ int lane_of_step(int step, int k)
{
    const unsigned long long m = __ballot(step == k);
    return wave_uniform(__ffsll((long long)m) - 1);
}

template <int N> struct RegK {
    double a[N];
    int step;
};

template <int N> __device__ __forceinline__ void regk_load(RegK<N> &R, const double *K, int lda)
{
    const int lane = lane_id(), row = lane < N ? lane : 0;
#pragma unroll
    for (int j = 0; j < N; ++j) R.a[j] = K[row * lda + j];
    R.step = N;
}
template <int N> __device__ __forceinline__ void regk_store(const RegK<N> &R, double *K, int lda, int *steps)
{
    const int lane = lane_id();
    if (lane < N) {
#pragma unroll
        for (int j = 0; j < N; ++j) K[lane * lda + j] = R.a[j];
        steps[lane] = R.step;
    }
}
template <int N> __device__ __forceinline__ void regk_reload(RegK<N> &R, const double *K, int lda, const int *steps)
{
    const int lane = lane_id(), row = lane < N ? lane : 0;
#pragma unroll
    for (int j = 0; j < N; ++j) R.a[j] = K[row * lda + j];
    R.step = lane < N ? steps[row] : -1;
}

template <int N> __device__ __forceinline__ void regk_factor(RegK<N> &R)
{
    const int lane = lane_id();
    if (lane >= N) R.step = -1;   // not a row
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const bool unused = R.step == N;
        unsigned key = 0u;
        if (unused) {
            // |a_ik| as float (monotone, clamped to the float range), low 6 bits replaced by 63 - lane
            const float vf = (float)fmin(fabs(R.a[k]), 3.0e38);
            key = ((__float_as_uint(vf) >> 1) & ~63u) | (unsigned)(63 - lane) | 0x80000000u;
        }
        key = wave_max_u32(key);
        const int p = wave_uniform(63 - (int)(key & 63u));
        if (lane == p) R.step = k;
        const double inv = 1.0 / wave_bcast(R.a[k], p);
        const bool upd = R.step == N;
        const double l = upd ? R.a[k] * inv : 0.0;   // rows that already served as pivot take a zero multiplier:
        if (upd) R.a[k] = l;                         // a - 0 * p == a exactly, and the update needs no select
        if (lane == p) R.a[k] = inv;
#pragma unroll
        for (int j = k + 1; j < N; ++j) {
            const double pj = wave_bcast(R.a[j], p);
            R.a[j] -= l * pj;
        }
    }
}

// Solve K x = b; lane i passes b_i and receives x_i (lanes >= N: 0).
template <int N> __device__ __forceinline__ double regk_solve(const RegK<N> &R, double x)
{
    const int lane = lane_id();
#pragma unroll
    for (int k = 0; k < N; ++k) {   // forward: replay the row operations on the right-hand side
        const int p = lane_of_step(R.step, k);
        const double bk = wave_bcast(x, p);
        if (R.step > k) x -= R.a[k] * bk;
    }
    double res = 0.0;
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {   // backward: U is spread over the pivot rows
        const int p = lane_of_step(R.step, k);
        const double xk = wave_bcast(x, p) * wave_bcast(R.a[k], p);
        if (lane == k) res = xk;
        if (R.step >= 0 && R.step < k) x -= R.a[k] * xk;
    }
    return res;
}

}  // namespace dss
