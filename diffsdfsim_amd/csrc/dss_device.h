// dss_device.h -- the single include through which kernels see the HIP device API.
// Product builds (hipcc, gfx950) get <hip/hip_runtime.h>.  tests/emu compiles the same sources
// with g++ -DDSS_EMU to debug kernel logic on the CPU-only build container (tests/emu/README.md);
// that build is test infrastructure and is never loaded by the product.
#pragma once
#if defined(DSS_EMU)
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#define DSS_DYN_LDS(type, name) extern __shared__ __align__(16) type name[]
#endif
