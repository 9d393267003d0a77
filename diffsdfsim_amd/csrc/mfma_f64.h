// mfma_f64.h -- the fp64 matrix-core tile used by the GEMM-shaped pieces of the path (gfx950).
//   D(16x16) += A(16x4) B(4x16):  lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15];
//   it holds D[row = (l >> 4) + 4 r][col = l & 15] in accumulator register r (r = 0..3).
#pragma once
#include "dss_device.h"

namespace dss {

#if defined(DSS_EMU)
typedef struct { double x, y, z, w; } acc4;
#else
typedef double acc4 __attribute__((ext_vector_type(4)));
#endif

__device__ inline acc4 mfma_f64_16x16x4(double a, double b, acc4 c)
{
#if defined(DSS_EMU)
    return dss_emu_mfma_f64_16x16x4(a, b, c);
#else
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
#endif
}
__device__ inline double &acc_comp(acc4 &v, int i)
{
#if defined(DSS_EMU)
    return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
#else
    return reinterpret_cast<double *>(&v)[i];
#endif
}

}  // namespace dss
