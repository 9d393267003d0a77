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
// min / max of a double over the wavefront with the GCN DPP reduction ladder (row_shr 1,2,3 | shr 4 | shr 8 |
// bcast15 | bcast31, total in lane 63): exact, order-independent, and a ~6x shorter dependent chain than six
// ds_bpermute shuffles, which matters with one wavefront per SIMD
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

}  // namespace dss
