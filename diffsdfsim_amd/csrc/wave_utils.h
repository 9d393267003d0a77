// wave_utils.h -- 64-lane wavefront helpers shared by the gfx950 kernels.
#pragma once
#include "dss_device.h"

namespace dss {

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, WAVE));
    return v;
}

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, WAVE));
    return v;
}

// arg-max of v with the LOWEST index winning ties (LAPACK idamax convention).
__device__ __forceinline__ void wave_argmax(double &v, int &idx)
{
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) {
        double ov = __shfl_xor(v, o, WAVE);
        int oi = __shfl_xor(idx, o, WAVE);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
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

}  // namespace dss
