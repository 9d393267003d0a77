// hip_emu.h -- TEST-ONLY functional emulation of the small HIP device subset our kernels use.
//
// The build container has no GPU.  To debug kernel *logic* on the CPU, tests/emu compiles the
// very same .hip sources with g++ -DDSS_EMU against this header: every thread of a block is a
// ucontext fiber, fibers run round-robin and switch at __syncthreads()/__shfl*, which gives the
// lock-step semantics a 64-lane wavefront has.  Nothing here is linked into, or reachable from,
// the product library (diffsdfsim_amd/csrc/libdiffsdfsim_hip.so); see tests/emu/README.md.
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__
#define __launch_bounds__(...)
#define __align__(n)
#define __restrict__
#define __shared__ static

typedef void *hipStream_t;
typedef int hipError_t;
constexpr int hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }   // (LDS is host memory here)
inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
typedef void *hipEvent_t;
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }

namespace dss_emu {
struct Fiber {
    ucontext_t ctx;
    std::vector<char> stack;
    bool done = false;
};
struct State {
    int bar_count = 0, bar_gen = 0, alive = 0;   // block barrier: arrivals, generation, fibers still running
    std::vector<Fiber> fibers;
    ucontext_t sched;
    int cur = 0;
    std::vector<char> dyn_lds;
    std::vector<uint64_t> slots;
    std::function<void()> body;
};
inline State &st() { static State s; return s; }
inline emu_uint3 &tidx() { static emu_uint3 v; return v; }
inline emu_uint3 &bidx() { static emu_uint3 v; return v; }
inline dim3 &bdim() { static dim3 v; return v; }
inline dim3 &gdim() { static dim3 v; return v; }

inline void yield()
{
    State &s = st();
    swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}
inline void trampoline()
{
    State &s = st();
    s.body();
    s.fibers[s.cur].done = true;
    --s.alive;
    if (s.alive > 0 && s.bar_count == s.alive) { s.bar_count = 0; ++s.bar_gen; }   // the others were waiting for this one
    swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}
inline void run_block(unsigned nthreads)
{
    State &s = st();
    s.fibers.clear();
    s.fibers.resize(nthreads);
    s.bar_count = 0; s.bar_gen = 0; s.alive = (int)nthreads;
    s.slots.assign(nthreads, 0);
    for (unsigned t = 0; t < nthreads; ++t) {
        Fiber &f = s.fibers[t];
        f.stack.resize(256 * 1024);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack.data();
        f.ctx.uc_stack.ss_size = f.stack.size();
        f.ctx.uc_link = &s.sched;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
    }
    bool alive = true;
    while (alive) {
        alive = false;
        for (unsigned t = 0; t < nthreads; ++t) {
            if (s.fibers[t].done) continue;
            s.cur = (int)t;
            tidx().x = t;
            swapcontext(&s.sched, &s.fibers[t].ctx);
            alive = true;
        }
    }
}
template <class F> inline void launch(dim3 grid, dim3 block, size_t lds_bytes, F &&f)
{
    State &s = st();
    gdim() = grid;
    bdim() = block;
    for (unsigned b = 0; b < grid.x; ++b) {
        bidx().x = b; bidx().y = 0; bidx().z = 0;
        s.dyn_lds.assign(lds_bytes + 64, 0);
        s.body = f;
        run_block(block.x);
    }
}
template <class T> inline T shfl_from(T v, int src)
{
    static_assert(sizeof(T) <= 8, "emu shuffle width");
    State &s = st();
    uint64_t raw = 0;
    std::memcpy(&raw, &v, sizeof(T));
    s.slots[s.cur] = raw;
    yield();
    int wave_base = (s.cur / 64) * 64;
    uint64_t got = s.slots[wave_base + (src & 63)];
    yield();
    T out;
    std::memcpy(&out, &got, sizeof(T));
    return out;
}
}  // namespace dss_emu

#define threadIdx (dss_emu::tidx())
#define blockIdx (dss_emu::bidx())
#define blockDim (dss_emu::bdim())
#define gridDim (dss_emu::gdim())

// a true workgroup barrier (wavefronts of one workgroup may run different code between barriers): wait until every fiber
// that is still running has arrived
inline void __syncthreads()
{
    dss_emu::State &s = dss_emu::st();
    const int gen = s.bar_gen;
    if (++s.bar_count == s.alive) { s.bar_count = 0; ++s.bar_gen; }
    else while (s.bar_gen == gen) dss_emu::yield();
    dss_emu::yield();
}
template <class T> inline T __shfl_xor(T v, int mask, int = 64) { return dss_emu::shfl_from(v, (dss_emu::st().cur & 63) ^ mask); }
template <class T> inline T __shfl(T v, int src, int = 64) { return dss_emu::shfl_from(v, src); }
template <class T> inline T __shfl_up(T v, int d, int = 64)
{
    int l = dss_emu::st().cur & 63;
    return dss_emu::shfl_from(v, l - d >= 0 ? l - d : l);
}
template <class T> inline T __shfl_down(T v, int d, int = 64)
{
    int l = dss_emu::st().cur & 63;
    return dss_emu::shfl_from(v, l + d < 64 ? l + d : l);
}
// block-wide predicate reductions: every fiber deposits its predicate, all read after the switch
inline int emu_block_reduce(int pred, int mode)
{
    dss_emu::State &s = dss_emu::st();
    s.slots[s.cur] = (uint64_t)(pred != 0);
    __syncthreads();
    int acc = (mode == 1) ? 1 : 0;
    for (size_t i = 0; i < s.slots.size(); ++i) {
        if (s.fibers[i].done) continue;
        int v = (int)s.slots[i];
        if (mode == 0) acc |= v; else if (mode == 1) acc &= v; else acc += v;
    }
    __syncthreads();
    return acc;
}
inline int __syncthreads_or(int p) { return emu_block_reduce(p, 0); }
inline int __syncthreads_and(int p) { return emu_block_reduce(p, 1); }
inline int __syncthreads_count(int p) { return emu_block_reduce(p, 2); }
inline unsigned long long __ballot(int pred)
{
    dss_emu::State &s = dss_emu::st();
    s.slots[s.cur] = (uint64_t)(pred != 0);
    dss_emu::yield();
    unsigned long long m = 0;
    int base = (s.cur / 64) * 64;
    for (int i = 0; i < 64 && base + i < (int)s.slots.size(); ++i) m |= (unsigned long long)(s.slots[base + i] & 1) << i;
    dss_emu::yield();
    return m;
}
inline long long wall_clock64() { return 0; }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline unsigned __float_as_uint(float f) { unsigned r; std::memcpy(&r, &f, 4); return r; }
inline long long __double_as_longlong(double d) { long long r; std::memcpy(&r, &d, 8); return r; }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline double atomicAdd(double *p, double v) { double o = *p; *p = o + v; return o; }
inline int atomicAdd(int *p, int v) { int o = *p; *p = o + v; return o; }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
inline int atomicOr(int *p, int v) { int o = *p; *p = o | v; return o; }
inline unsigned atomicOr(unsigned *p, unsigned v) { unsigned o = *p; *p = o | v; return o; }
inline int atomicMax(int *p, int v) { int o = *p; if (v > o) *p = v; return o; }

// v_mfma_f64_16x16x4_f64 as documented for gfx950 (cdna_hip_programming.md section 3): lane l supplies
// A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; it receives D[row = (l>>4) + 4 r][col = l&15] in register r.
struct dss_emu_acc4 { double x, y, z, w; };
template <class ACC> inline ACC dss_emu_mfma_f64_16x16x4(double a, double b, ACC c)
{
    dss_emu::State &s = dss_emu::st();
    static std::vector<double> Aall(1024), Ball(1024);   // one 64-entry slab per wavefront of the block
    const int l = s.cur & 63;
    double *A = Aall.data() + (s.cur & ~63), *B = Ball.data() + (s.cur & ~63);
    A[l] = a; B[l] = b;
    dss_emu::yield();
    double *cv = reinterpret_cast<double *>(&c);
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) + 4 * r, col = l & 15;
        double acc = cv[r];
        for (int k = 0; k < 4; ++k) acc += A[16 * k + row] * B[16 * k + col];
        cv[r] = acc;
    }
    dss_emu::yield();
    return c;
}

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    dss_emu::launch((grid), (block), (lds), [=]() { kernel(__VA_ARGS__); })

// dynamic LDS: the kernels declare it through DSS_DYN_LDS(type, name)
#define DSS_DYN_LDS(type, name) type *name = reinterpret_cast<type *>(dss_emu::st().dyn_lds.data())
