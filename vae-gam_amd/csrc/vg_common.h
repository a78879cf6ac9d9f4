// vg_common.h -- shared definitions for the VAE-GAM HIP kernels (gfx950 / MI355X).
//
// The kernels are written once, in HIP, for CDNA4 (64-wide wavefronts, LDS tiles).  For
// debugging index arithmetic without a GPU, tests/emu/ compiles the same sources with g++
// and -DVG_EMU: tests/emu/hip_emu.h then supplies threadIdx/blockIdx/__syncthreads/__shfl
// on host threads.  That build is a test harness only; the product library is the hipcc one.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef VG_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#define VG_DYN_SMEM(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
template <typename K, typename... A>
static inline void vg_launch(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t s, A... args) {
    hipLaunchKernelGGL(kernel, grid, block, shmem, s, args...);
}
#endif

#define VG_WAVE 64

enum vg_status {
    VG_OK = 0,
    VG_ERR_ARG = 1,        // bad shape / null pointer / unsupported configuration
    VG_ERR_LAUNCH = 2,     // HIP reported a launch error
    VG_ERR_UNSUPPORTED = 3 // layer geometry has no compiled kernel instance
};

void vg_set_error(const char* fmt, ...);
int vg_check_launch(const char* what);

// prologue applied to an activation tensor when a consumer loads it (activations are stored
// as PRE-activation values; the consumer applies ReLU and the batch-norm affine on load)
struct VgPrologue {
    const float* scale;   // [groups][C] or nullptr
    const float* shift;   // [groups][C] or nullptr
    int relu;             // apply max(x,0) first
    int per_group;        // samples per statistic group (n / per_group = group); ignored if scale==nullptr
};

__host__ __device__ static inline int vg_cdiv(int a, int b) { return (a + b - 1) / b; }
