// kkt_reg.h -- register-resident LU of the reduced KKT matrix (n <= 64): lane i owns row i.
//
// The LDS version (factor_K / solve_K in lcp_contact.hip) spends its time in latency chains: a shuffle
// reduction for the pivot, three barriers and dependent LDS read-modify-writes per elimination step.
// Here the matrix lives in VGPRs with compile-time indices (N is a template parameter, every loop is
// fully unrolled), the pivot row is broadcast with v_readlane through SGPRs and rows never move:
// partial pivoting is implicit (`step` = elimination step at which a row served as pivot).
//   row i, before it is a pivot:   a[k] (k < current step) = multiplier l_ik
//   row i, once pivot at step s:   a[j] (j > s) = U_sj,  a[s] = 1 / U_ss
// Pivot choice: largest |a_ik| among unused rows, compared on the top 58 bits (ties -> lowest lane).
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
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) {
        const unsigned long long w = __shfl_xor(v, o, WAVE);
        v = w > v ? w : v;
    }
    return v;
}
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
        unsigned long long key = 0ull;
        if (unused) {
            const double v = fabs(R.a[k]);
            key = ((unsigned long long)__double_as_longlong(v) & ~63ull) | (unsigned long long)(63 - lane);
            key |= 1ull << 63;   // any unused row beats "no row" even when its entry is +0
        }
        key = wave_max_u64(key);
        const int p = wave_uniform(63 - (int)(key & 63ull));
        if (lane == p) R.step = k;
        const double inv = 1.0 / wave_bcast(R.a[k], p);
        const bool upd = R.step == N;
        const double l = R.a[k] * inv;
        if (upd) R.a[k] = l;
        if (lane == p) R.a[k] = inv;
#pragma unroll
        for (int j = k + 1; j < N; ++j) {
            const double pj = wave_bcast(R.a[j], p);
            if (upd) R.a[j] -= l * pj;
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
