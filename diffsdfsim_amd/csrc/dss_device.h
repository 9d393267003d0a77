// dss_device.h -- the single include through which kernels see the HIP device API.
// Product builds (hipcc, gfx950) get <hip/hip_runtime.h>.  tests/emu compiles the same sources
// with g++ -DDSS_EMU to debug kernel logic on the CPU-only build container (tests/emu/README.md);
// that build is test infrastructure and is never loaded by the product.
#pragma once
#if defined(DSS_EMU)
#include "hip_emu.h"
// lanes of one emulated wavefront meet at every switch point
inline void dss_wave_sync() { dss_emu::yield(); }
inline int dss_uniform(int x) { return x; }
inline int dss_opaque(int x) { return x; }
inline double dss_rcp(double x) { return 1.0 / x; }
inline double dss_uniform(double x) { return x; }
#else
#include <hip/hip_runtime.h>
#define DSS_DYN_LDS(type, name) extern __shared__ __align__(16) type name[]
// LDS written by one lane is visible to the other lanes of the SAME wavefront afterwards (no s_barrier: the LDS
// queue of a wavefront is in order; the fence only stops the compiler from moving accesses across)
__device__ __forceinline__ void dss_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// the value, hidden from the optimiser (stops loop-invariant code motion of everything computed from it)
__device__ __forceinline__ int dss_opaque(int x) { asm volatile("" : "+v"(x)); return x; }
// 1 / x to within an ulp or two: v_rcp_f64 and two Newton steps (for arguments well inside the normal range)
__device__ __forceinline__ double dss_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
    return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
}
// a value every lane of the wavefront agrees on, moved to scalar registers
__device__ __forceinline__ int dss_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ double dss_uniform(double x)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
#endif

// A big by-value kernel argument (DssWorld, ~1 KB) whose address is handed to non-inlined device functions is copied by the
// compiler into every lane's private scratch memory (64 KB per wavefront of stores at kernel entry, and every later field
// access a scratch load).  Referring to it where it already lies -- the kernarg segment, readable by every lane -- avoids the
// copy.  `arg` must be the kernel's FIRST parameter (offset 0 of the segment).
#if defined(DSS_EMU)
#define DSS_KERNARG_REF(T, name, arg) const T &name = arg
#define DSS_KERNARG_REF_AT(T, name, arg, offset) const T &name = arg
#else
#define DSS_KERNARG_REF(T, name, arg) const T &name = *(const T *)__builtin_amdgcn_kernarg_segment_ptr(); (void)arg
// a later parameter: `offset` = its byte offset in the segment (parameters are laid out in order at their natural alignment)
#define DSS_KERNARG_REF_AT(T, name, arg, offset) \
    const T &name = *(const T *)((const char *)__builtin_amdgcn_kernarg_segment_ptr() + (offset)); (void)arg
#endif
