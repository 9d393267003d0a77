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

}  // namespace dss
