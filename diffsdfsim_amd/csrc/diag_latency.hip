// diag_latency.hip -- diagnostic micro-benchmarks (compiled to nothing unless -DDSS_DIAG; tools/latency.py).
// One wavefront per SIMD is how the LCP kernel runs, so the raw dependent-issue latencies of gfx950 matter.
#include "dss_device.h"
#if defined(DSS_DIAG) && !defined(DSS_EMU)
namespace {
__global__ void __launch_bounds__(64) lat_kernel(int mode, int n, double *sink, long long *out, const double *gbuf, const int *chase)
{
    __shared__ double lds[1024];
    __shared__ int lchase[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) { lds[i] = 1.0 + i * 1e-9; lchase[i] = (i * 37 + 11) & 1023; }
    __syncthreads();
    double a = 1.0 + lane * 1e-9, b = 1.0000001, c = 1e-9, a2 = a, a3 = a, a4 = a;
    int idx = lane;
    const long long w0 = wall_clock64(), c0 = clock64();
    if (mode == 0) for (int i = 0; i < n; ++i) a = fma(a, b, c);                                   // dependent fp64 FMA
    else if (mode == 1) for (int i = 0; i < n; ++i) { a = fma(a, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c); a4 = fma(a4, b, c); }
    else if (mode == 2) for (int i = 0; i < n; ++i) {                                               // readlane x2 + fma
        const int lo = __builtin_amdgcn_readlane(__double2loint(a2), 5), hi = __builtin_amdgcn_readlane(__double2hiint(a2), 5);
        a = fma(a, __hiloint2double(hi, lo), c); a2 += 1e-12;
    }
    else if (mode == 3) for (int i = 0; i < n; ++i) { idx = lchase[idx]; }                          // dependent LDS reads
    else if (mode == 4) for (int i = 0; i < n; ++i) { idx = chase[idx]; }                           // dependent global (L2) reads
    else if (mode == 5) for (int i = 0; i < n; ++i) a = a / b;                                      // dependent fp64 divide
    else if (mode == 6) for (int i = 0; i < n; ++i) a = sqrt(a) + 1.0;                              // dependent fp64 sqrt
    else if (mode == 7) for (int i = 0; i < n; ++i) { a = __shfl_xor(a, 1 + (i & 31), 64) + c; }    // dependent shuffle
    const long long c1 = clock64(), w1 = wall_clock64();
    if (lane == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = w1 - w0; }
    sink[blockIdx.x * 64 + lane] = a + a2 + a3 + a4 + idx;
}
}  // namespace
extern "C" void dss_diag_latency(int mode, int n, int grid, double *sink, long long *out, const double *gbuf, const int *chase, void *stream)
{
    hipLaunchKernelGGL(lat_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, mode, n, sink, out, gbuf, chase);
}
#endif
