// kkt_reg.h -- register-resident LU of the reduced KKT matrix (n <= 64): lane i owns row i.
//
// The LDS version (factor_K / solve_K in lcp_contact.hip) spends its time in latency chains: a shuffle
// reduction for the pivot, three barriers and dependent LDS read-modify-writes per elimination step.
// Here the matrix lives in VGPRs with compile-time indices (N is a template parameter, every loop is
// fully unrolled), the pivot row is broadcast with v_readlane through SGPRs and rows never move.
//   row i:   a[k] (k < i) = multiplier l_ik,   a[i] = 1 / U_ii,   a[j] (j > i) = U_ij
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
template <int N> struct RegK {
    double a[N];
};

// ---- elimination in natural order (no pivot search) ---------------------------------------------------------------
// The reduced KKT matrix is [[H, A^T],[A, 0]] with H = Q + sum P C P^T symmetric positive definite and A of full row
// rank: every leading pivot of H is positive and the Schur complement -A H^-1 A^T that is left for the last rows is
// negative definite, so Gaussian elimination in natural order never meets a zero pivot and is backward stable for
// the definite blocks.  Without the search the pivot lane is a compile-time constant (v_readlane with an immediate),
// which removes the DPP ladder, the key arithmetic and the ballots of the pivoted version from the serial chain.
template <int N> __device__ __forceinline__ void regk_factor_natural(RegK<N> &R)
{
    const int lane = lane_id();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double inv = 1.0 / wave_bcast(R.a[k], k);
        const bool upd = lane > k && lane < N;
        const double l = upd ? R.a[k] * inv : 0.0;
        if (upd) R.a[k] = l;
        if (lane == k) R.a[k] = inv;
#pragma unroll
        for (int j = k + 1; j < N; ++j) R.a[j] -= l * wave_bcast(R.a[j], k);
    }
}
template <int N> __device__ __forceinline__ double regk_solve_natural(const RegK<N> &R, double x)
{
    const int lane = lane_id();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double bk = wave_bcast(x, k);
        if (lane > k && lane < N) x -= R.a[k] * bk;
    }
    double res = 0.0;
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {
        const double xk = wave_bcast(x, k) * wave_bcast(R.a[k], k);
        if (lane == k) res = xk;
        if (lane < k) x -= R.a[k] * xk;
    }
    return res;
}

// The same on the leading M x M part of the register array (M <= N): used when a body is pinned by identity
// equality rows, which removes its six rows and the six multiplier rows from the system to factor (lcp_contact.hip).
template <int N, int M> __device__ __forceinline__ void regk_factor_lead(RegK<N> &R)
{
    const int lane = lane_id();
#pragma unroll
    for (int k = 0; k < M; ++k) {
        const double inv = 1.0 / wave_bcast(R.a[k], k);
        const bool upd = lane > k && lane < M;
        const double l = upd ? R.a[k] * inv : 0.0;
        if (upd) R.a[k] = l;
        if (lane == k) R.a[k] = inv;
#pragma unroll
        for (int j = k + 1; j < M; ++j) R.a[j] -= l * wave_bcast(R.a[j], k);
    }
}
// The same for a BLOCK-TRIDIAGONAL leading part (6 x 6 blocks: every contact joins bodies that are neighbours in the body order,
// or a body and the pinned one -- a stack): elimination without pivoting keeps that shape, so row k only changes the columns of
// its own and the next block.  The updates left out are `a[j] -= l * 0.0`: the factors are bit-identical, at 357 instead of 861
// broadcast + multiply-add pairs for M = 42.
template <int N, int M> __device__ __forceinline__ void regk_factor_lead_tri(RegK<N> &R)
{
    const int lane = lane_id();
#pragma unroll
    for (int k = 0; k < M; ++k) {
        const double inv = 1.0 / wave_bcast(R.a[k], k);
        const bool upd = lane > k && lane < M;
        const double l = upd ? R.a[k] * inv : 0.0;
        if (upd) R.a[k] = l;
        if (lane == k) R.a[k] = inv;
#pragma unroll
        for (int j = k + 1; j < M; ++j)
            if (j < 6 * (k / 6 + 2)) R.a[j] -= l * wave_bcast(R.a[j], k);
    }
}
template <int N, int M> __device__ __forceinline__ double regk_solve_lead(const RegK<N> &R, double x)
{
    const int lane = lane_id();
#pragma unroll
    for (int k = 0; k < M; ++k) {
        const double bk = wave_bcast(x, k);
        if (lane > k && lane < M) x -= R.a[k] * bk;
    }
    double res = 0.0;
#pragma unroll
    for (int k = M - 1; k >= 0; --k) {
        const double xk = wave_bcast(x, k) * wave_bcast(R.a[k], k);
        if (lane == k) res = xk;
        if (lane < k) x -= R.a[k] * xk;
    }
    return res;
}

}  // namespace dss
