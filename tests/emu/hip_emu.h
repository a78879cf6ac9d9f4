// hip_emu.h -- TEST HARNESS ONLY.  Minimal host emulation of the HIP execution model so the
// kernel sources under vae-gam_amd/csrc can be compiled with g++ (-DVG_EMU) and run on tiny shapes.
// The threads of a block are fibers (ucontext) on one OS thread: __syncthreads / __shfl / MFMA are
// barriers that switch fibers, so divergent-barrier bugs deadlock here instead of corrupting a GPU;
// blocks are spread over a few OS threads (`__shared__` is thread_local static, one copy per runner).
// Never linked into the product library.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

using std::min; using std::max;
struct float4 { float x, y, z, w; };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef void* hipStream_t;
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __restrict__ __restrict

struct EmuBar { int count = 0, arrived = 0; unsigned gen = 0; };
struct EmuWave { float fbuf[64]; double dbuf[64]; int lanes; EmuBar bar; };
struct EmuBlock {
    EmuBar bar;
    std::vector<EmuWave> waves;
    std::vector<char> dyn;
};
extern thread_local EmuBlock* g_emu_block;
extern thread_local dim3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;
extern thread_local int emu_tid;
extern std::mutex g_emu_atomic_mu;

void emu_yield();
static inline void emu_wait(EmuBar& b) {
    if (++b.arrived == b.count) { b.arrived = 0; ++b.gen; return; }
    const unsigned g = b.gen;
    while (b.gen == g) emu_yield();
}

#define VG_DYN_SMEM(type, name) type* name = reinterpret_cast<type*>(g_emu_block->dyn.data())

static inline void __syncthreads() { emu_wait(g_emu_block->bar); }
static inline float __shfl_down(float v, int d) {
    EmuWave& w = g_emu_block->waves[emu_tid / 64]; int lane = emu_tid % 64;
    w.fbuf[lane] = v; emu_wait(w.bar);
    float r = (lane + d < w.lanes) ? w.fbuf[lane + d] : v; emu_wait(w.bar); return r;
}
static inline double __shfl_down(double v, int d) {
    EmuWave& w = g_emu_block->waves[emu_tid / 64]; int lane = emu_tid % 64;
    w.dbuf[lane] = v; emu_wait(w.bar);
    double r = (lane + d < w.lanes) ? w.dbuf[lane + d] : v; emu_wait(w.bar); return r;
}
static inline float __shfl_xor(float v, int m) {
    EmuWave& w = g_emu_block->waves[emu_tid / 64]; int lane = emu_tid % 64;
    w.fbuf[lane] = v; emu_wait(w.bar);
    float r = ((lane ^ m) < w.lanes) ? w.fbuf[lane ^ m] : v; emu_wait(w.bar); return r;
}
static inline float atomicAdd(float* p, float v) { std::lock_guard<std::mutex> g(g_emu_atomic_mu); float o = *p; *p = o + v; return o; }
static inline double atomicAdd(double* p, double v) { std::lock_guard<std::mutex> g(g_emu_atomic_mu); double o = *p; *p = o + v; return o; }
static inline int hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(int) { return "emu"; }

void emu_run(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);

template <typename K, typename... A>
static inline void vg_launch(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t, A... args) {
    emu_run(grid, block, shmem, [=]() { kernel(args...); });
}
