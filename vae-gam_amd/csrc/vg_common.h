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
    (void)hipGetLastError();        // drop any stale error another library left on this thread; vg_check_launch reads ours
    hipLaunchKernelGGL(kernel, grid, block, shmem, s, args...);
}
#endif

#define VG_WAVE 64

// blocks of `kernel` (block size, dynamic LDS bytes) that fit one CU at once: a persistent grid is sized to exactly the
// resident blocks (a larger grid runs its surplus blocks as a second, mostly idle round)
#ifdef VG_EMU
static inline int vg_blocks_per_cu(const void*, int, size_t shmem) { int n = (int)((160 * 1024) / (shmem ? shmem : 1)); return n < 1 ? 1 : (n > 8 ? 8 : n); }
#else
static inline int vg_blocks_per_cu(const void* kernel, int block, size_t shmem) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, block, shmem) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 1; }
    return n;
}
#endif

// wave index inside the block as a wave-uniform (scalar) value: row/tile decoding that depends only
// on it then runs on the scalar ALU
#ifdef VG_EMU
static inline int vg_wave_id() { return threadIdx.x / VG_WAVE; }
#else
__device__ __forceinline__ int vg_wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x / VG_WAVE)); }
#endif

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

// LDS-DMA: 4 bytes per lane from a per-lane global address straight into LDS at (wave-uniform base +
// lane*4), no VGPR round trip (global_load_lds_dword).  vg_dma_wait() retires this wave's DMAs; a
// barrier must follow before other waves read the bytes.
#ifdef VG_EMU
static inline void vg_dma4(const float* gsrc, float* lds_row_base) { lds_row_base[emu_tid % 64] = *gsrc; }
static inline void vg_dma_wait() {}
// lanes are host fibers here: the wave-wide completion a real vmcnt(0) gives (a lane may then read what its neighbours copied) is a
// wave barrier; call it from wave-uniform code only
static inline void vg_dma_wait_wave() { emu_wait(g_emu_block->waves[emu_tid / 64].bar); }
#else
#ifdef VG_DMA_ASM
// The same instruction issued from inline assembly: the compiler does not see an LDS write in flight, so it does not put a
// `s_waitcnt vmcnt(0)` in front of the next LDS read of the wave (with the builtin it does, whatever the addresses: a copy issued ahead
// of a compute phase is then waited for at that phase's first read).  Every consumer waits explicitly (vg_dma_wait / vg_wait_vm + barrier).
__device__ __forceinline__ unsigned vg_lds_addr(const float* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)p);      // (the low half of a generic LDS address is the LDS byte offset)
}
__device__ __forceinline__ void vg_dma4(const float* gsrc, float* lds_row_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(gsrc), "s"(vg_lds_addr(lds_row_base)) : "memory", "m0");
}
#else
__device__ __forceinline__ void vg_dma4(const float* gsrc, float* lds_row_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_row_base, 4, 0, 0);
}
#endif
__device__ __forceinline__ void vg_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// same instruction; the name says that the caller then reads LDS bytes OTHER lanes of its wave copied (a wave waits as a whole)
__device__ __forceinline__ void vg_dma_wait_wave() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif

// Wave-cooperative copy of n contiguous floats -> LDS: the full 64-float DMA instructions carry no per-lane predicate
// (only the tail does), so each costs a scalar add or two.  src_lane = source + lane.
#ifdef VG_EMU
static inline void vg_dma_span(const float* src_lane, float* dst, int n, int lane) {
    for (int o = 0; o + lane < n; o += 64) dst[o + lane] = src_lane[o];
}
#else
__device__ __forceinline__ void vg_dma_span(const float* src_lane, float* dst, int n, int lane) {
    for (; n >= 64; n -= 64, src_lane += 64, dst += 64) vg_dma4(src_lane, dst);     // running pointers: the tail reuses them
    if (lane < n) vg_dma4(src_lane, dst);
}
#endif

// 16 bytes per lane (global_load_lds_dwordx4): source and LDS destination 16-byte aligned, n a multiple of 4.  src_lane = source + 4*lane.
#ifdef VG_EMU
static inline void vg_dma_span16(const float* src_lane, float* dst, int n, int lane) {
    for (int o = 0; o + 4 * lane < n; o += 256) for (int e = 0; e < 4; ++e) dst[o + 4 * lane + e] = src_lane[o + e];
}
#else
#ifdef VG_DMA_ASM
__device__ __forceinline__ void vg_dma16(const float* gsrc, float* lds_row_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(vg_lds_addr(lds_row_base)) : "memory", "m0");
}
#else
__device__ __forceinline__ void vg_dma16(const float* gsrc, float* lds_row_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_row_base, 16, 0, 0);
}
#endif
__device__ __forceinline__ void vg_dma_span16(const float* src_lane, float* dst, int n, int lane) {
    for (; n >= 256; n -= 256, src_lane += 256, dst += 256) vg_dma16(src_lane, dst);
    if (4 * lane < n) vg_dma16(src_lane, dst);
}
#endif

// n contiguous floats, global -> LDS, by the `nw` waves of a block together (nw = 1, wave = 0: by one wave alone).  The source needs
// only 4-byte alignment (measured: tools/micro/dma16_align.hip -- global_load_lds_dwordx4 takes any dword-aligned address), so
// everything from the first 16-byte boundary of the LDS destination on moves as 1 KB wave-instructions (a quarter of the
// instructions of the dword form, whose issue was 5-25 % of a wave's cycles in the conv / weight-gradient kernels); the up to 3 floats
// in front of that boundary and behind the last whole 16 bytes go as dword copies.  Wave w takes the whole 1 KB pieces w, w + nw, ...
#ifdef VG_EMU
static inline void vg_dma_block(const float* src, float* dst, int n, int wave, int nw, int lane) {
    for (int i = wave * 64 + lane; i < n; i += nw * 64) dst[i] = src[i];
}
#else
__device__ __forceinline__ void vg_dma_block(const float* src, float* dst, int n, int wave, int nw, int lane) {
    int h = (int)((4u - (((unsigned)(uintptr_t)dst >> 2) & 3u)) & 3u);
    if (h > n) h = n;
    if (wave == 0 && lane < h) vg_dma4(src + lane, dst);
    const int nb = (n - h) >> 8;
    const float* s_ = src + h + 4 * lane; float* d_ = dst + h;
    for (int k = wave; k < nb; k += nw) vg_dma16(s_ + (k << 8), d_ + (k << 8));
    if (wave == nb % nw) {
        const int r0 = h + (nb << 8), r4 = (n - r0) >> 2;
        if (lane < r4) vg_dma16(src + r0 + 4 * lane, dst + r0);
        const int t0 = r0 + (r4 << 2);
        if (lane < n - t0) vg_dma4(src + t0 + lane, dst + t0);
    }
}
#endif

// Wait until at most n (wave-uniform, a multiple of 4 up to 32) of this wave's vector-memory operations are outstanding: they retire in
// issue order, so everything issued BEFORE the last n -- e.g. an LDS-DMA copy issued ahead of n stores -- has completed, while the stores
// keep draining.  n outside the table waits for everything.
#ifdef VG_EMU
static inline void vg_wait_vm(int) { emu_wait(g_emu_block->waves[emu_tid / 64].bar); }
static inline bool vg_any(bool p) {
    EmuWave& w = g_emu_block->waves[emu_tid / 64]; const int lane = emu_tid % 64;
    w.fbuf[lane] = p ? 1.f : 0.f; emu_wait(w.bar);
    bool r = false; for (int l = 0; l < w.lanes; ++l) r = r || (w.fbuf[l] != 0.f);
    emu_wait(w.bar); return r;
}
#else
__device__ __forceinline__ void vg_wait_vm(int n) {
    switch (n) {
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
        case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
__device__ __forceinline__ bool vg_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
#endif

// Read-only kernel arguments that must stay on the SCALAR load path (weights indexed by wave-uniform values): loads through
// the constant address space are invariant by definition, so the compiler keeps them s_load even in kernels that also
// store to LDS/global (where its no-clobber analysis otherwise gives up and turns them into per-lane VMEM loads + VGPRs).
#ifdef VG_EMU
typedef const float* vg_cptr;
#define VG_CPTR(p) (p)
#else
typedef const __attribute__((address_space(4))) float* vg_cptr;
#define VG_CPTR(p) ((vg_cptr)(p))
#endif

// max(x, lo) as ONE v_max_f32: fmaxf() makes the compiler quiet both inputs first (two extra v_max per call); the operands
// here are finite activations and lo is 0 or -inf, so IEEE NaN handling is irrelevant.
#ifdef VG_EMU
static inline float vg_max(float x, float lo) { return x > lo ? x : lo; }
#else
__device__ __forceinline__ float vg_max(float x, float lo) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(lo)); return r; }
#endif

// fp32-input MFMA, exact fp32 (v_mfma_f32_16x16x4_f32): D(16x16) = A(16x4) * B(4x16) + C.
// lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; it owns D[row = (l>>4)*4 + r][col = l&15], r = 0..3.
struct vg_f32x4 { float v[4]; };
#ifdef VG_EMU
static inline void vg_mfma16(float a, float b, vg_f32x4& acc) {
    EmuWave& w = g_emu_block->waves[emu_tid / 64]; const int lane = emu_tid % 64;
    w.fbuf[lane] = a; emu_wait(w.bar);
    float arow[4][4];                                  // A[row][k] for this lane's 4 rows
    for (int r = 0; r < 4; ++r) for (int k = 0; k < 4; ++k) arow[r][k] = w.fbuf[k * 16 + (lane >> 4) * 4 + r];
    emu_wait(w.bar);
    w.fbuf[lane] = b; emu_wait(w.bar);
    for (int r = 0; r < 4; ++r) { float s = acc.v[r]; for (int k = 0; k < 4; ++k) s = fmaf(arow[r][k], w.fbuf[k * 16 + (lane & 15)], s); acc.v[r] = s; }
    emu_wait(w.bar);
}
#else
typedef float vg_hw_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void vg_mfma16(float a, float b, vg_f32x4& acc) {
    vg_hw_f32x4 c = {acc.v[0], acc.v[1], acc.v[2], acc.v[3]};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    acc.v[0] = c[0]; acc.v[1] = c[1]; acc.v[2] = c[2]; acc.v[3] = c[3];
}
#endif

// two adjacent floats as ONE 8-byte store at a 4-byte-aligned address (rows of odd pitch start on odd dwords): gfx950 runs global
// accesses in unaligned mode, so the under-aligned dwordx2 is legal; two strided dword stores per lane instead leave every
// 64-byte line half-written by each instruction (measured as 1.9x write amplification on the stride-2 transposed conv)
struct __attribute__((packed, aligned(4))) vg_f2u { float a, b; };
__device__ __forceinline__ void vg_store2(float* p, float a, float b) { vg_f2u v; v.a = a; v.b = b; *reinterpret_cast<vg_f2u*>(p) = v; }

// Sum over the 64 lanes of a wave, returned in every lane.  On the GPU: six DPP adds inside the vector ALU (row shifts 1, 2, 4, 8, then
// the two row broadcasts; the total lands in lane 63 and is read back as a scalar) instead of six ds_bpermute round trips through the
// LDS pipe per __shfl_down reduction -- the fused GAM/ELBO backward does 8 of them per sample and wave.
#ifdef VG_EMU
static inline float vg_wave_sum(float v) {
    EmuWave& w = g_emu_block->waves[emu_tid / 64]; const int lane = emu_tid % 64;
    w.fbuf[lane] = v; emu_wait(w.bar);
    float r = 0.f; for (int l = 0; l < w.lanes; ++l) r += w.fbuf[l];
    emu_wait(w.bar); return r;
}
#else
__device__ __forceinline__ float vg_wave_sum(float v) {
#define VG_DPP_ADD(ctrl, rmask) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
    VG_DPP_ADD(0x111, 0xf); VG_DPP_ADD(0x112, 0xf); VG_DPP_ADD(0x114, 0xf); VG_DPP_ADD(0x118, 0xf);      // row_shr:1 2 4 8: inclusive scan in each row of 16
    VG_DPP_ADD(0x142, 0xa);                                                                               // row_bcast:15 into rows 1, 3
    VG_DPP_ADD(0x143, 0xc);                                                                               // row_bcast:31 into rows 2, 3
#undef VG_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#endif

// a value the optimiser may not see through, pinned to a vector register (it would turn 0/1 factors and all-ones/zero masks back into
// boolean predicates, i.e. scalar register pairs)
#ifdef VG_EMU
static inline float vg_opaque(float v) { return v; }
static inline unsigned vg_opaque(unsigned v) { return v; }
#else
__device__ __forceinline__ float vg_opaque(float v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ unsigned vg_opaque(unsigned v) { asm volatile("" : "+v"(v)); return v; }
#endif

// value with its bits and-ed by an all-ones / all-zero mask: an exact select that keeps its condition in a VECTOR register
__host__ __device__ static inline float vg_and(float v, unsigned m) {
    union { float f; unsigned u; } t; t.f = v; t.u &= m; return t.f;
}

__host__ __device__ static inline int vg_cdiv(int a, int b) { return (a + b - 1) / b; }
