// hip_emu.h -- TEST HARNESS ONLY.  Minimal host emulation of the HIP execution model so the
// kernel sources under vae-gam_amd/csrc can be compiled with g++ (-DVG_EMU) and run under
// AddressSanitizer on tiny shapes.  One OS thread per GPU thread of a block, blocks run one
// after another; __syncthreads/__shfl are real barriers, so divergent-barrier bugs deadlock
// here instead of corrupting a GPU.  Never linked into the product library.
#pragma once
#include <barrier>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <algorithm>

using std::min; using std::max;
struct float4 { float x, y, z, w; };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef void* hipStream_t;
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__ __restrict

struct EmuWave { float fbuf[64]; double dbuf[64]; int lanes; std::unique_ptr<std::barrier<>> bar; };
struct EmuBlock {
    std::unique_ptr<std::barrier<>> bar;
    std::vector<EmuWave> waves;
    std::vector<char> dyn;
};
extern EmuBlock* g_emu_block;
extern thread_local dim3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
extern thread_local int emu_tid;
extern std::mutex g_emu_atomic_mu;

#define VG_DYN_SMEM(type, name) type* name = reinterpret_cast<type*>(g_emu_block->dyn.data())

static inline void __syncthreads() { g_emu_block->bar->arrive_and_wait(); }
static inline float __shfl_down(float v, int d) {
    EmuWave& w = g_emu_block->waves[emu_tid / 64]; int lane = emu_tid % 64;
    w.fbuf[lane] = v; w.bar->arrive_and_wait();
    float r = (lane + d < w.lanes) ? w.fbuf[lane + d] : v; w.bar->arrive_and_wait(); return r;
}
static inline double __shfl_down(double v, int d) {
    EmuWave& w = g_emu_block->waves[emu_tid / 64]; int lane = emu_tid % 64;
    w.dbuf[lane] = v; w.bar->arrive_and_wait();
    double r = (lane + d < w.lanes) ? w.dbuf[lane + d] : v; w.bar->arrive_and_wait(); return r;
}
static inline float __shfl_xor(float v, int m) {
    EmuWave& w = g_emu_block->waves[emu_tid / 64]; int lane = emu_tid % 64;
    w.fbuf[lane] = v; w.bar->arrive_and_wait();
    float r = ((lane ^ m) < w.lanes) ? w.fbuf[lane ^ m] : v; w.bar->arrive_and_wait(); return r;
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
