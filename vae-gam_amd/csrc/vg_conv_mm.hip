// vg_conv_mm.hip -- table-driven implicit-GEMM convolution on the fp32 matrix cores (gfx950, v_mfma_f32_16x16x4_f32: exact fp32).
//
// One kernel for the conv / transposed-conv forward passes and data gradients of vae_reg_GP.py:189-215 whose channel counts make
// them real contractions (8 or 16 output channels): D[row][pos] += sum_kappa A[row][kappa] * B[kappa][pos] with
//   rows  = 16 = (16 / CO) "replicas" x CO output channels.  With 8 output channels the upper 8 rows compute the NEXT output
//           column from the same B operand (Toeplitz pair: their weights are shifted by one column / one stride), so no matrix
//           row idles;
//   pos   = 16 consecutive positions of a block's position grid (PD planes x all rows x all columns of the layer);
//   kappa = (input channel, window offset): the window offsets of every k-step come from a TABLE built on the host
//           (ops.mm_plan): strided correlations (window = kernel, plus the replica shift), stride-2 transposed convolutions in
//           gather form (one "class" per output parity (rd, rh), window = the taps of that parity), padded or not, share it.
// A operand: a [class][ci][k-step][64 lanes] image of the layer's weights (zeros where a row has no weight for that offset),
//   gathered from the flat parameter buffer once per step (vg_gather_f32), copied to LDS once per (persistent) block.
// B operand: ONE ds_read_b32 per MFMA at (position base + table offset) out of the input planes the block needs, which are
//   whole planes of the tensor = one contiguous span per channel, copied flat into LDS by LDS-DMA (no halo cells, no per-row
//   address arithmetic), double-buffered over (sample, channel chunk) behind the matrix work.  Zero padding / tile overhang:
//   a per-lane bit mask per accumulator tile (bit s = "the element k-step s reads exists"), built once per block.
//   The producer's ReLU / batch-norm affine is applied once per staged element, in place, by the thread that copied it.
// Blocks are persistent over samples: masks, position offsets, the A image and the offset table are set up once.
#include "vg_common.h"
#include <stdlib.h>
#include <stdint.h>
#include "../../include/vaegam.h"

#ifdef VG_STAMP
// Diagnostic build only (tools/stamp_bench.py): per-wave cycle sums of the phases of a unit, read back through vg_stamp_read.
__device__ unsigned long long vg_stamp_out[1024 * 16 * 8];
#define VG_STAMP_T(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define VG_STAMP_ADD(i) do { unsigned long long t_; VG_STAMP_T(t_); st_sum[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define VG_STAMP_ADD(i) do {} while (0)
#endif

namespace {

constexpr int MM_MAXQ = 4;                 // (workgroups are MM_W = 8 or 4 wavefronts: a template parameter of the kernel)
constexpr int MM_ZPAD = 64;                // zero floats at the head of every channel slot: where B operands that do not exist are read from

struct MmParams {
    vg_mm_desc d;
    int bps, nsplit;                       // blocks per sample, sample splits (grid = bps * nsplit)
    int ksbase[MM_MAXQ], a_off[MM_MAXQ];   // first table row / first A-image float of class q
    int kstot, aimg_floats;
    int a_res;                             // 1: the whole A image stays in LDS; 0: the slices of the running channel chunk are staged with it
    int CHP, buf_floats;                   // channel pitch, floats of the input buffer
    int nhb, PHB, h0, LH, LPH;             // row slabs per plane, position rows per slab, first staged row relative to ph0*shi, staged rows, LDS plane pitch
    int tau_off, in_off;                   // LDS float offsets
    int has_pro;
};

struct alignas(16) mm_f4 { float v[4]; };   // one ds_read_b128 / ds_write_b128

template <int V> struct mm_int { static constexpr int value = V; };
// f(mm_int<n>) for the run-time n in [1, MAXN]
template <int MAXN, typename F>
__device__ __forceinline__ void mm_dispatch(int n, F&& f) {
    if constexpr (MAXN >= 1) {
        if (n == MAXN) { f(mm_int<MAXN>{}); return; }
        mm_dispatch<MAXN - 1>(n, f);
    }
}
// f(mm_int<0>) ... f(mm_int<N-1>)
template <int N, int I = 0, typename F>
__device__ __forceinline__ void mm_static_for(F&& f) {
    if constexpr (I < N) { f(mm_int<I>{}); mm_static_for<N, I + 1>(f); }
}

// K0..K3: k-steps per input channel of class 0..3 as compile-time constants (0 = read them from the descriptor: generic, slower).
// With them the k-step loop unrolls into straight-line code, the LDS offset of every (tile, step) operand lives in a register and
// every LDS read of a channel is issued ahead of the matrix instructions that consume it (a run-time loop pays two dependent LDS
// latencies per step).  Classes (NQ = 4: the output parities of a stride-2 transposed conv) are processed one after the other on
// the same staged input, each stored as soon as it is finished: its stores drain behind the next class's matrix work.
//
// Round 3 (in-kernel stamps, tools/diag/stamp_bench.py): inside the matrix phase the MFMA pipe is 75-87 % busy, but it idles through the
// phases every wave of a block enters together -- input wait / prologue / barrier / epilogue stores were 32 % (convt3 fwd) to 60 %
// (convt4 fwd, convt3's data gradient) of a wave's cycles.  So: (i) OCC = waves per SIMD the register budget is held to (4 = 128
// VGPRs: TWO blocks per CU whose phases interleave); (ii) DB = 0 stages the input single-buffered (half the LDS: what lets two
// blocks fit; the co-resident block covers the exposed copy); (iii) the epilogue works from per-tile output offsets and validity
// bits set up once per block, stores through a uniform base + 32-bit lane offset, and -- data gradients -- fetches the ReLU mask
// of a sample BEFORE that sample's last matrix phase instead of between its stores (that wait was 48 % of convt3's data gradient);
// (iv) a wave copies AND post-processes whole channel spans (16-byte LDS accesses), statistics are flushed per group run.
template <int MM_W, int NQ, int TPC, int K0, int K1, int K2, int K3, bool DB, int OCC, bool MASKED>
__global__ void __launch_bounds__(MM_W * VG_WAVE, OCC)
conv_mm_k(const float* __restrict__ x, const float* __restrict__ a_img, const int* __restrict__ tau, const int* __restrict__ dlt,
          const float* __restrict__ bias, const float* __restrict__ in_scale, const float* __restrict__ in_shift,
          const float* __restrict__ mask_src, float* __restrict__ y, double* __restrict__ stats_part, int stats_pg, int stats_relu,
          MmParams p) {
    VG_DYN_SMEM(float, lds);
    constexpr int MM_T = MM_W * VG_WAVE;
    const vg_mm_desc& d = p.d;
    float* Al = lds;
    int* Tl = reinterpret_cast<int*>(lds + p.tau_off);
    float* In = lds + p.in_off;
    const int tid = threadIdx.x, lane = tid % VG_WAVE, wave = vg_wave_id();
    const int kk = lane >> 4, jl = lane & 15;
    // (an XCD-aware order -- the slabs of one sample on one XCD, so that they share the halo planes in its L2 -- measured neutral to 15 % slower)
    const int b = blockIdx.x % p.bps, split = blockIdx.x / p.bps;
    const int IHW = d.IH * d.IW, OHW = d.OH * d.OW;
    const int CI = d.CI, CO = d.CO;
    constexpr bool STATIC_K = K0 > 0;
    constexpr int KSUM = K0 + K1 + K2 + K3;

    for (int i = tid; i < p.aimg_floats; i += MM_T) Al[i] = a_img[i];
    if (!STATIC_K) for (int i = tid; i < p.kstot * 64; i += MM_T) {
        const int* e = dlt + ((i >> 6) * 4 + ((i & 63) >> 4)) * 3;
        Tl[i] = e[0] * p.LPH + e[1] * d.IW + e[2];
    }
    for (int i = tid; i < (DB ? 2 : 1) * d.cc * MM_ZPAD; i += MM_T) In[(i / MM_ZPAD) * p.CHP + (i % MM_ZPAD)] = 0.f;  // the zero pads (never written again)

    // ---- geometry of this block (the same for every sample it visits): slab b = (plane slab bd, row slab bh) of the position grid
    const int bd = b / p.nhb, bh = b - bd * p.nhb;
    const int pd0 = bd * d.PD, ph0 = bh * p.PHB;
    const int npd = min(d.PD, d.PDT - pd0), nph = min(p.PHB, d.PH - ph0);
    const int npos = npd * nph * d.PW;
    const int ntiles = (npos + 15) / 16;
    const int dlo = pd0 * d.sdi + d.d0;                                  // first staged input plane (may lie outside the tensor)
    const int pl_lo = max(dlo, 0), pl_hi = min(dlo + d.LD, d.ID);
    const int npl = max(pl_hi - pl_lo, 0);
    const int rlo = ph0 * d.shi + p.h0;                                  // first staged input row of every plane (may lie outside it)
    const int r_lo = max(rlo, 0), r_hi = min(rlo + p.LH, d.IH);
    const int LPH = p.LPH;
    // staged rows of a plane are one contiguous span; if they are ALL rows of the plane and the LDS plane pitch is the tensor's, the
    // planes of a channel are one span as well
    const bool flat = (r_lo == 0 && r_hi == d.IH && LPH == IHW);
    const int span_fl = flat ? npl * IHW : max(r_hi - r_lo, 0) * d.IW;   // floats per span
    const int nspan_c = flat ? 1 : npl;                                  // spans per channel
    // LDS float offset (past the zero pad) of the first staged element.  A plane's slot begins with its first STAGED row (r_lo: rows
    // outside the tensor take no space -- their operands point at the zero pad) and LPH is a multiple of 4 (host), so every span starts
    // on a 16-byte boundary; a span's last 16-byte group may reach up to 3 floats past its end -- into the slack the host leaves
    // behind every plane (LPH >= rows*IW + 4), never into another span
    const int dst0 = (pl_lo - dlo) * LPH;
    const size_t vol = (size_t)IHW * d.ID, ovol = (size_t)OHW * d.OD;
    // rows 4*kk .. 4*kk+3 of a tile: replica and first channel are lane constants
    const int rho_l = (CO == 8) ? (kk >> 1) : 0;
    const int co_l = (CO == 8) ? ((kk & 1) * 4) : kk * 4;
    // Per accumulator tile and k-step: WHERE this lane's B operand sits inside a channel slot -- the element (pd*sdi + dd, ph*shi + dh,
    // pw*swi + dw) of the staged planes, or, where that element does not exist (zero padding, tile overhang, lanes past the last
    // position), the slot's zero pad.  Float offsets from the slot base; with compile-time step counts they live in registers
    // (pt), so a matrix instruction costs one address add + one ds_read_b32 and no select.
    // Per tile: ob = offset of the lane's first output element of class 0 inside the sample (channel co_l, replica rho_l); bit
    // q*TPC + i of vbits = "tile i of class q writes an element that exists".
    int ob[TPC];
    unsigned vbits = 0;
    int pt[TPC][STATIC_K ? KSUM : 1];
    int posBase[TPC]; unsigned vm[NQ][TPC];                              // run-time step counts only
#pragma unroll
    for (int i = 0; i < TPC; ++i) {
        const int pf = (i * MM_W + wave) * 16 + jl;
        const bool pv = pf < npos;
        const int pfc = pv ? pf : 0;
        const int pdl = pfc / (nph * d.PW), r2 = pfc - pdl * (nph * d.PW);
        const int phl = r2 / d.PW, pw = r2 - phl * d.PW;
        const int ph = ph0 + phl;
        const int obd = (pd0 + pdl) * d.sdo, obh = ph * d.sho, obw = pw * d.swo + rho_l;
        ob[i] = obd * OHW + obh * d.OW + obw + co_l * (int)ovol;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int od = obd + d.od0[q], oh = obh + d.oh0[q], ow = obw + d.ow0[q];
            const bool ok = pv && od >= 0 && od < d.OD && oh >= 0 && oh < d.OH && ow >= 0 && ow < d.OW;
            vbits |= (ok ? 1u : 0u) << (q * TPC + i);
        }
        const int pb = (pdl * d.sdi - d.d0) * LPH + (ph * d.shi - r_lo) * d.IW + pw * d.swi;
        posBase[i] = pb;
        if constexpr (STATIC_K) {
#pragma unroll
            for (int sg = 0; sg < KSUM; ++sg) {                          // sg = table row (classes back to back)
                const int* e = dlt + (sg * 4 + kk) * 3;
                const int id = (pd0 + pdl) * d.sdi + e[0], ih = ph * d.shi + e[1], iw = pw * d.swi + e[2];
                const bool ok = pv && id >= 0 && id < d.ID && ih >= 0 && ih < d.IH && iw >= 0 && iw < d.IW;
                pt[i][sg] = ok ? MM_ZPAD + pb + e[0] * LPH + e[1] * d.IW + e[2] : 0;      // (the host's tau assumes whole planes)
            }
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                unsigned m = 0;
                for (int s = 0; s < d.ks[q]; ++s) {
                    const int* e = dlt + ((p.ksbase[q] + s) * 4 + kk) * 3;
                    const int id = (pd0 + pdl) * d.sdi + e[0], ih = ph * d.shi + e[1], iw = pw * d.swi + e[2];
                    const bool ok = pv && id >= 0 && id < d.ID && ih >= 0 && ih < d.IH && iw >= 0 && iw < d.IW;
                    m |= (ok ? 1u : 0u) << s;
                }
                vm[q][i] = m;
            }
        }
    }
    const float lo = d.relu_in ? 0.f : -__builtin_inff();
    const int nchunks = (CI + d.cc - 1) / d.cc;
    const int nsamp = (d.N - split + p.nsplit - 1) / p.nsplit;          // samples this block visits: split, split + nsplit, ...
    const int units = nsamp * nchunks;

    vg_f32x4 acc[TPC];
    float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};
    float mreg[MASKED ? TPC : 1][4];                                     // ReLU mask source of the sample in flight (data gradients)
    float bias_l[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias_l[r] = bias ? bias[co_l + r] : 0.f;
    int nact = 0;                                                        // tiles this wave owns (wave-uniform)
#pragma unroll
    for (int i = 0; i < TPC; ++i) nact += (i * MM_W + wave < ntiles) ? 1 : 0;

    // stores a wave issues for class q (4 per tile with a lane that writes): what a counted vmcnt lets fly past the next input wait
    int nst[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        int c4 = 0;
#pragma unroll
        for (int i = 0; i < TPC; ++i) c4 += (vg_any(((vbits >> (q * TPC + i)) & 1u) != 0) && i < nact) ? 4 : 0;
        nst[q] = c4;
    }

    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TPC; ++i) { acc[i].v[0] = 0.f; acc[i].v[1] = 0.f; acc[i].v[2] = 0.f; acc[i].v[3] = 0.f; }
    };
    // fetch the mask source of sample n's outputs (class 0) into registers; consumed by store_class
    auto load_mask = [&](int n) __attribute__((always_inline)) {
        const char* mn = reinterpret_cast<const char*>(mask_src + (size_t)n * CO * ovol);
#pragma unroll
        for (int i = 0; i < TPC; ++i) {
            const bool okt = i < nact && ((vbits >> i) & 1u);
            const unsigned off = okt ? 4u * (unsigned)(ob[i] + d.od0[0] * OHW + d.oh0[0] * d.OW + d.ow0[0]) : 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                mreg[i][r] = okt ? *reinterpret_cast<const float*>(mn + (size_t)r * ovol * 4 + off) : 0.f;
        }
    };
    // store class q of sample n from the accumulators: lane owns position jl of each tile, rows 4*kk .. 4*kk+3
    auto store_class = [&](int n, int q) __attribute__((always_inline)) {
        char* yn = reinterpret_cast<char*>(y + (size_t)n * CO * ovol);
        const char* mn = reinterpret_cast<const char*>(mask_src + (size_t)n * CO * ovol);
        const int cq = d.od0[q] * OHW + d.oh0[q] * d.OW + d.ow0[q];       // wave-uniform
#pragma unroll
        for (int i = 0; i < TPC; ++i) {
            if (i >= nact || !((vbits >> (q * TPC + i)) & 1u)) continue;
            const unsigned off = 4u * (unsigned)(ob[i] + cq);             // bytes inside the sample: uniform base + 32-bit lane offset
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i].v[r] + bias_l[r];
                if constexpr (MASKED) {
                    if constexpr (NQ == 1) v = (mreg[i][r] > 0.f) ? v : 0.f;                       // fetched before the matrix phase
                    else v = (*reinterpret_cast<const float*>(mn + (size_t)r * ovol * 4 + off) > 0.f) ? v : 0.f;
                }
#if defined(VG_ABLATE_STORE) && VG_ABLATE_STORE == 1
                if (v == 1.2345e-30f) *reinterpret_cast<float*>(yn + (size_t)r * ovol * 4 + off) = v;        // diagnostic: (almost) no store
#elif defined(VG_ABLATE_STORE) && VG_ABLATE_STORE == 2
                *reinterpret_cast<float*>(yn + (size_t)r * ovol * 4 + (((off >> 8) << 8) + 4u * lane)) = v;    // diagnostic: one aligned 256-byte run per instruction
#elif defined(VG_ABLATE_STORE) && VG_ABLATE_STORE == 4
                if (r == 0 || v == 1.2345e-30f) *reinterpret_cast<float*>(yn + (size_t)r * ovol * 4 + off) = v;            // diagnostic: a quarter of the stores
#elif defined(VG_ABLATE_STORE) && VG_ABLATE_STORE == 5
                *reinterpret_cast<float*>(yn + off) = v;                                                         // diagnostic: all four stores to the first one's address
#elif defined(VG_ABLATE_STORE) && VG_ABLATE_STORE == 3
                __builtin_nontemporal_store(v, reinterpret_cast<float*>(yn + (size_t)r * ovol * 4 + off));          // diagnostic: nt stores
#else
                *reinterpret_cast<float*>(yn + (size_t)r * ovol * 4 + off) = v;
#endif
                if (stats_part) {
                    const float h = stats_relu ? vg_max(v, 0.f) : v;
                    st_s[r] += h; st_q[r] = fmaf(h, h, st_q[r]);
                }
            }
        }
    };
    auto flush_stats = [&](int n) __attribute__((always_inline)) {
        // per wavefront and run of samples of one group: [sum, sum of squares] of relu?(y) per channel, in the slot of the run's last
        // sample (the other slots of the run stay zero), laid out for vg_bn_stats_from_parts
        const int g = n / stats_pg;
        const size_t chunks = (size_t)stats_pg * p.bps * MM_W;
        const size_t chunkid = ((size_t)(n % stats_pg) * p.bps + b) * MM_W + wave;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sa = st_s[r], sb = st_q[r];
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) { sa += __shfl_xor(sa, off); sb += __shfl_xor(sb, off); }
            if (CO == 8) { sa += __shfl_xor(sa, 32); sb += __shfl_xor(sb, 32); }   // the two replicas of a channel
            if (jl == 0 && (CO == 16 || kk < 2)) {
                double* dst = stats_part + (((size_t)g * CO + co_l + r) * chunks + chunkid) * 2;
                dst[0] = (double)sa; dst[1] = (double)sb;
            }
            st_s[r] = 0.f; st_q[r] = 0.f;
        }
    };

    // Input of unit u (sample, channel chunk): per channel the staged rows of every staged plane -- one contiguous span per plane, or one
    // per channel where whole planes at the tensor's own pitch are staged (`flat`) -- by LDS-DMA.  A span is cut into `parts` pieces of
    // whole 16-byte groups and every (channel, span, piece) belongs to ONE wave, which copies it and -- after its own vmcnt wait --
    // applies the producer's ReLU / batch-norm affine to it in place: no barrier between copy and prologue.
    const int nsp = d.cc * nspan_c;
    const int parts = (nsp >= MM_W) ? 1 : (MM_W + nsp - 1) / nsp;
    // non-flat: every span starts on a 16-byte boundary (above).  flat (tensor pitch, one span per channel): it starts `head` floats past
    // one; the floats of its first / last 16-byte group outside the span lie inside the channel slot, past the zero pad, and no operand
    // offset points at them
    const int head = dst0 & 3;
    const int n4 = (head + span_fl + 3) >> 2;                            // 16-byte groups that cover a span
    const int n4p = (n4 + parts - 1) / parts;
    auto stage = [&](int u) __attribute__((always_inline)) {
        const int n = split + (u / nchunks) * p.nsplit, c0 = (u % nchunks) * d.cc;
        const int cc = min(d.cc, CI - c0);
        float* buf = In + (DB ? (u & 1) : 0) * p.buf_floats;
        const float* src0 = x + ((size_t)n * CI + c0) * vol + (size_t)pl_lo * IHW + (size_t)r_lo * d.IW;
        for (int it = wave; it < cc * nspan_c * parts; it += MM_W) {
            const int sp = it / parts, part = it - sp * parts;
            const int c = sp / nspan_c, pl = sp - c * nspan_c;
            const int g0 = part * n4p, g1 = min(g0 + n4p, n4);
            const int f0 = max(g0 * 4 - head, 0), f1 = min(g1 * 4 - head, span_fl);     // floats [f0, f1) of the span
            if (f1 > f0) vg_dma_block(src0 + (size_t)c * vol + (size_t)pl * IHW + f0, buf + c * p.CHP + MM_ZPAD + dst0 + pl * LPH + f0, f1 - f0, 0, 1, lane);
        }
    };
    auto prologue = [&](int u) __attribute__((always_inline)) {
        const int n = split + (u / nchunks) * p.nsplit, c0 = (u % nchunks) * d.cc;
        const int cc = min(d.cc, CI - c0);
        float* buf = In + (DB ? (u & 1) : 0) * p.buf_floats;
        const int g_aff = (in_scale != nullptr) ? n / d.per_group : 0;
        for (int it = wave; it < cc * nspan_c * parts; it += MM_W) {
            const int sp = it / parts, part = it - sp * parts;
            const int c = sp / nspan_c, pl = sp - c * nspan_c;
            float sc = 1.f, sh = 0.f;
            if (in_scale != nullptr) { sc = in_scale[g_aff * CI + c0 + c]; sh = in_shift[g_aff * CI + c0 + c]; }
            float* sp0 = buf + c * p.CHP + MM_ZPAD + dst0 + pl * LPH - head;    // the 16-byte boundary at / in front of the span's first float
            const int g0 = part * n4p, g1 = min((part + 1) * n4p, n4);
#ifdef VG_EMU
            for (int g = g0 + lane; g < g1; g += VG_WAVE) {
                float* t = sp0 + g * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = fmaf(vg_max(t[e], lo), sc, sh);
            }
#else
            // LDS accesses as inline asm: in front of a ds_read the compiler can see, its wait-count pass puts s_waitcnt vmcnt(0) while an
            // LDS-DMA is outstanding -- which also waits for every store of the previous unit (the counted wait above exists to avoid that)
            for (int g = g0 + lane; g < g1; g += VG_WAVE) {
                const unsigned a = (unsigned)(uintptr_t)(sp0 + g * 4);
                vg_hw_f32x4 t;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(a) : "memory");
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = fmaf(vg_max(t[e], lo), sc, sh);
                asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(t) : "memory");
            }
#endif
        }
#ifndef VG_EMU
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the asm writes are invisible to the compiler's own wait in front of the barrier
#endif
    };
    __syncthreads();
#ifdef VG_STAMP
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last;
    VG_STAMP_T(st_last);
#endif
    if (units > 0) stage(0);
    int g_run = -1, n_run = 0;                                           // statistics: group / last sample of the running partial sums
    int pend = 0;                                                        // single-buffered: stores issued AFTER the copy of the unit about to start
    for (int u = 0; u < units; ++u) {
        const int n = split + (u / nchunks) * p.nsplit, chunk = u % nchunks, c0 = chunk * d.cc;
        const int cc = min(d.cc, CI - c0);
        float* cur = In + (DB ? (u & 1) : 0) * p.buf_floats;
        VG_STAMP_ADD(7);
        if (DB) vg_dma_wait_wave();                                      // this wave's share of unit u has landed
        else vg_wait_vm(pend);                                           // ... while the stores issued behind its copy keep draining
        VG_STAMP_ADD(0);
        if (p.has_pro) prologue(u);
        VG_STAMP_ADD(1);
        __syncthreads();                                                 // unit u complete in LDS; every wave is past unit u-1
        VG_STAMP_ADD(2);
        if (DB && u + 1 < units) stage(u + 1);                           // into the buffer unit u-1 used: lands behind this unit's matrix work
        if constexpr (NQ == 1 && MASKED) { if (chunk == nchunks - 1) load_mask(n); }   // lands behind the matrix phase
        VG_STAMP_ADD(3);
        // ---- matrix work.  The body is instantiated per number of tiles THIS wave owns (wave-uniform): no per-tile branches inside.
        auto class_work = [&](auto qc, auto ntc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            constexpr int NT = decltype(ntc)::value;
            for (int c = 0; c < cc; ++c) {
                const float* curc = cur + c * p.CHP;
                if constexpr (STATIC_K) {
                    constexpr int KSQ = q == 0 ? K0 : q == 1 ? K1 : q == 2 ? K2 : K3;
                    constexpr int SG0 = q == 0 ? 0 : q == 1 ? K0 : q == 2 ? K0 + K1 : K0 + K1 + K2;
                    const float* Aq = Al + p.a_off[q] + (c0 + c) * KSQ * 64 + lane;
                    // Hand-scheduled software pipeline over the k-steps: the operands of step s+1 are requested before the matrix
                    // instructions of step s issue, and the wait in front of them is COUNTED (lgkmcnt(NT+1): everything but the NT+1 reads
                    // just issued has landed -- LDS reads return in order).  The compiler only ever emits lgkmcnt(0) in this loop, i.e.
                    // it waits for the reads it has just issued: every step then exposes a full LDS latency (matrix pipe busy 35 %; with
                    // its own two-step batching 24 %).  So the reads are inline asm, which its wait-count pass does not track.
#ifdef VG_EMU
#pragma unroll
                    for (int s = 0; s < KSQ; ++s) {
                        const float aw = Aq[s * 64];
#pragma unroll
                        for (int i = 0; i < NT; ++i) vg_mfma16(aw, curc[pt[i][SG0 + s]], acc[i]);
                    }
#else
                    const unsigned cb = (unsigned)(uintptr_t)curc, ab = (unsigned)(uintptr_t)Aq;
                    float a2[2], b2[2][NT];
                    asm volatile("ds_read_b32 %0, %1" : "=v"(a2[0]) : "v"(ab));
#pragma unroll
                    for (int i = 0; i < NT; ++i) asm volatile("ds_read_b32 %0, %1" : "=v"(b2[0][i]) : "v"(cb + 4u * (unsigned)pt[i][SG0]));
#pragma unroll
                    for (int s = 0; s < KSQ; ++s) {
                        if (s + 1 < KSQ) {
                            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a2[(s + 1) & 1]) : "v"(ab), "n"((s + 1) * 256));
#pragma unroll
                            for (int i = 0; i < NT; ++i)
                                asm volatile("ds_read_b32 %0, %1" : "=v"(b2[(s + 1) & 1][i]) : "v"(cb + 4u * (unsigned)pt[i][SG0 + s + 1]));
                            asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(NT + 1) : "memory");
                        } else {
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < NT; ++i) vg_mfma16(a2[s & 1], b2[s & 1][i], acc[i]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#endif
                } else {
                    const int ksq = d.ks[q];
                    const float* Aq = Al + p.a_off[q] + (c0 + c) * ksq * 64 + lane;
                    const int* Tq = Tl + p.ksbase[q] * 64 + lane;
                    for (int s = 0; s < ksq; ++s) {
                        const float aw = Aq[s * 64];
                        const int off = Tq[s * 64] + MM_ZPAD;
                        const unsigned bit = 1u << s;
                        float bv[NT];
#pragma unroll
                        for (int i = 0; i < NT; ++i) bv[i] = curc[(vm[q][i] & bit) ? posBase[i] + off : 0];
#pragma unroll
                        for (int i = 0; i < NT; ++i) vg_mfma16(aw, bv[i], acc[i]);
                    }
                }
            }
        };
        const int g_now = stats_part ? n / stats_pg : 0;
        if (stats_part && chunk == 0 && g_run >= 0 && g_now != g_run) flush_stats(n_run);   // the run of g_run's samples has ended
        if constexpr (NQ == 1) {
            if (chunk == 0) zero_acc();
            if (nact > 0) mm_dispatch<TPC>(nact, [&](auto ntc) __attribute__((always_inline)) { class_work(mm_int<0>{}, ntc); });
            VG_STAMP_ADD(4);
            pend = 0;
            if constexpr (!DB) {
                // single buffer: the next unit's copy goes out as soon as every wave has read its last operand, AHEAD of this unit's stores
                if (u + 1 < units) { __syncthreads(); VG_STAMP_ADD(2); stage(u + 1); VG_STAMP_ADD(3); }
                if (chunk == nchunks - 1) pend = nst[0];
            }
            if (chunk == nchunks - 1) store_class(n, 0);
            VG_STAMP_ADD(5);
        } else {
            // all channels are resident (the host plans one chunk per sample for multi-class layers)
            mm_static_for<NQ>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value;
                zero_acc();
                VG_STAMP_ADD(5);
                if (nact > 0) mm_dispatch<TPC>(nact, [&](auto ntc) __attribute__((always_inline)) { class_work(qc, ntc); });
                VG_STAMP_ADD(4);
                if constexpr (!DB && q == NQ - 1) {                      // as above: copy of the next unit first, the last class's stores behind it
                    if (u + 1 < units) { __syncthreads(); VG_STAMP_ADD(2); stage(u + 1); VG_STAMP_ADD(3); }
                    pend = nst[q];
                }
                store_class(n, q);
            });
            VG_STAMP_ADD(5);
        }
        g_run = g_now; n_run = n;
    }
    if (stats_part && g_run >= 0) flush_stats(n_run);
    VG_STAMP_ADD(6);
#ifdef VG_STAMP
    if (lane == 0 && blockIdx.x < 1024)
        for (int i = 0; i < 8; ++i) vg_stamp_out[(blockIdx.x * 16 + wave) * 8 + i] = st_sum[i];
#endif
}

}  // namespace

#ifdef VG_STAMP
extern "C" int vg_stamp_read(unsigned long long* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(vg_stamp_out), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
#endif

static int mm_plan_params(const vg_mm_desc* d, MmParams* p, size_t* shmem, const char* who) {
    if (!d || d->N <= 0 || d->CI <= 0 || (d->CO != 8 && d->CO != 16) || d->nq < 1 || d->nq > MM_MAXQ || d->PD <= 0 || d->PDT <= 0 ||
        d->PH <= 0 || d->PW <= 0 || d->slack < 0 || d->PH > 1023 || d->PW > 1023 || d->PD > 1023 || d->cc <= 0 || d->LD <= 0 || d->tpc <= 0) {
        vg_set_error("%s: bad descriptor", who); return VG_ERR_ARG;
    }
    p->d = *d;
    int ns = 256 / p->bps; if (ns < 1) ns = 1; if (ns > d->N) ns = d->N;
    p->nsplit = ns;
    int row = 0, af = 0;
    for (int q = 0; q < d->nq; ++q) {
        if (d->ks[q] <= 0 || d->ks[q] > 32) { vg_set_error("%s: k-steps per channel must be 1..32 (class %d: %d)", who, q, d->ks[q]); return VG_ERR_ARG; }
        p->ksbase[q] = row; p->a_off[q] = af;
        row += d->ks[q]; af += d->CI * d->ks[q] * 64;
    }
    p->kstot = row; p->aimg_floats = af;
    if (d->nq > 1 && d->cc < d->CI) { vg_set_error("%s: multi-class plans need all input channels in one chunk", who); return VG_ERR_ARG; }
    const int IHW = d->IH * d->IW;
    const int W = d->waves;
    if (W != 4 && W != 8) { vg_set_error("%s: waves per workgroup must be 4 or 8 (got %d)", who, W); return VG_ERR_ARG; }
    if (d->hhi < d->hlo) { vg_set_error("%s: bad row-offset range [%d, %d]", who, d->hlo, d->hhi); return VG_ERR_ARG; }
    // row slabs: PHB position rows per block (0 / >= PH: whole planes); a slab stages the input rows its windows touch
    p->PHB = (d->PHB > 0 && d->PHB < d->PH) ? d->PHB : d->PH;
    p->nhb = vg_cdiv(d->PH, p->PHB);
    p->h0 = d->hlo;
    p->LH = (p->PHB - 1) * d->shi + (d->hhi - d->hlo) + 1;
    const int rows_max = p->LH < d->IH ? p->LH : d->IH;                  // rows outside the tensor take no LDS
    // whole planes keep the tensor's own pitch (one flat copy per channel where every row is staged); row slabs: the staged rows +
    // slack for the last 16-byte group of a plane's span, planes on 16-byte boundaries
    const bool whole = p->PHB == d->PH && d->hlo <= 0 && (d->PH - 1) * d->shi + d->hhi + 1 >= d->IH;    // every row of every staged plane: `flat` in the kernel
    p->LPH = whole ? IHW : ((rows_max * d->IW + 3) / 4) * 4 + 4;
    p->CHP = MM_ZPAD + ((d->LD * p->LPH + 63) / 64) * 64;                // [zero pad][LD planes]
    p->buf_floats = d->cc * p->CHP;
    p->a_res = 1;
    p->tau_off = ((p->aimg_floats + 63) / 64) * 64;
    p->in_off = p->tau_off + p->kstot * 64;
    p->bps = vg_cdiv(d->PDT, d->PD) * p->nhb;
    const size_t total = (size_t)p->in_off + (d->dbuf ? 2 : 1) * (size_t)p->buf_floats + 64;
    *shmem = total * sizeof(float);
    if (*shmem > 160 * 1024) { vg_set_error("%s: plan needs %zu bytes of LDS", who, *shmem); return VG_ERR_UNSUPPORTED; }
    if (d->PD * p->PHB * d->PW > d->tpc * W * 16) { vg_set_error("%s: %d positions per block exceed tpc=%d x %d waves", who, d->PD * p->PHB * d->PW, d->tpc, W); return VG_ERR_ARG; }
    return VG_OK;
}

extern "C" int64_t vg_conv_mm_stats_chunks(const vg_mm_desc* d, int32_t stats_per_group) {
    if (!d || stats_per_group <= 0 || d->PD <= 0 || d->PDT <= 0 || d->PH <= 0 || (d->waves != 4 && d->waves != 8)) return -1;
    const int PHB = (d->PHB > 0 && d->PHB < d->PH) ? d->PHB : d->PH;
    return (int64_t)stats_per_group * vg_cdiv(d->PDT, d->PD) * vg_cdiv(d->PH, PHB) * d->waves;
}

extern "C" int vg_conv_mm(const vg_mm_desc* d, const float* x, const float* a_img, const int32_t* tau, const int32_t* dlt,
                          const float* bias, const float* in_scale, const float* in_shift, const float* mask_src, float* y,
                          int32_t stats_per_group, int32_t stats_relu, double* stats_part, void* stream) {
    MmParams p; size_t shmem;
    int rc = mm_plan_params(d, &p, &shmem, "vg_conv_mm");
    if (rc) return rc;
    if (!x || !a_img || !tau || !dlt || !y) { vg_set_error("vg_conv_mm: null argument"); return VG_ERR_ARG; }
    if ((in_scale == nullptr) != (in_shift == nullptr) || (in_scale && d->per_group <= 0)) { vg_set_error("vg_conv_mm: in_scale/in_shift/per_group inconsistent"); return VG_ERR_ARG; }
    if (stats_part && (stats_per_group <= 0 || d->N % stats_per_group)) { vg_set_error("vg_conv_mm: bad statistics arguments"); return VG_ERR_ARG; }
    p.has_pro = (d->relu_in || in_scale) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    // persistent grid = the blocks that are resident at once (occupancy query of the chosen instance x 256 CUs), dealt over the
    // bps position slabs of a sample: every further block of a slab takes every nsplit-th sample
    const int threads = d->waves * VG_WAVE;
    auto launch = [&](auto kernel) {
        const int bpc = vg_blocks_per_cu((const void*)kernel, threads, shmem);
        int ns = (256 * bpc) / p.bps; if (ns < 1) ns = 1; if (ns > d->N) ns = d->N;
        p.nsplit = ns;
        vg_launch(kernel, dim3(p.bps * p.nsplit), dim3(threads), shmem, s, x, a_img, (const int*)tau, (const int*)dlt, bias, in_scale, in_shift,
                  mask_src, y, stats_part, (int)(stats_part ? stats_per_group : 1), (int)stats_relu, p);
    };
    const bool db = d->dbuf != 0, mk = mask_src != nullptr, w4 = d->waves == 4;
    // OCC: 4 waves per SIMD (128 registers: two 8-wave or four 4-wave blocks per CU) where the per-(tile, step) operand offsets leave room for it
#define MM_LAUNCH_W(W, NQ, TPC, K0, K1, K2, K3) { \
        constexpr int OCC_ = (TPC * (K0 + K1 + K2 + K3) + 10 * TPC <= 96 && K0 > 0) ? 4 : 2; \
        if (db) { if (mk) launch(conv_mm_k<W, NQ, TPC, K0, K1, K2, K3, true, OCC_, true>); else launch(conv_mm_k<W, NQ, TPC, K0, K1, K2, K3, true, OCC_, false>); } \
        else    { if (mk) launch(conv_mm_k<W, NQ, TPC, K0, K1, K2, K3, false, OCC_, true>); else launch(conv_mm_k<W, NQ, TPC, K0, K1, K2, K3, false, OCC_, false>); } }
#define MM_LAUNCH(NQ, TPC, K0, K1, K2, K3) { if (w4) MM_LAUNCH_W(4, NQ, TPC, K0, K1, K2, K3) else MM_LAUNCH_W(8, NQ, TPC, K0, K1, K2, K3) }
#define MM_TPC(NQ, K0, K1, K2, K3) \
    { if (d->tpc <= 3) MM_LAUNCH(NQ, 3, K0, K1, K2, K3) else if (d->tpc <= 4) MM_LAUNCH(NQ, 4, K0, K1, K2, K3) else if (d->tpc <= 5) MM_LAUNCH(NQ, 5, K0, K1, K2, K3) \
      else if (d->tpc <= 6) MM_LAUNCH(NQ, 6, K0, K1, K2, K3) else MM_LAUNCH(NQ, 8, K0, K1, K2, K3) }
    const int* k = d->ks;
    if (d->nq == 1 && d->tpc <= 8) {
        if (k[0] == 7) MM_TPC(1, 7, 0, 0, 0)
        else if (k[0] == 9) MM_TPC(1, 9, 0, 0, 0)
        else if (k[0] == 12) MM_TPC(1, 12, 0, 0, 0)
        else if (k[0] == 19 && d->tpc <= 3) MM_LAUNCH(1, 3, 19, 0, 0, 0)
        else MM_TPC(1, 0, 0, 0, 0)
    } else if (d->nq == 4 && d->tpc <= 4) {
        if (k[0] == 3 && k[1] == 2 && k[2] == 2 && k[3] == 1) MM_LAUNCH(4, 4, 3, 2, 2, 1)
        else if (k[0] == 2 && k[1] == 1 && k[2] == 1 && k[3] == 1) MM_LAUNCH(4, 4, 2, 1, 1, 1)
        else if (k[0] == 2 && k[1] == 2 && k[2] == 2 && k[3] == 2) MM_LAUNCH(4, 4, 2, 2, 2, 2)          // 4x4x4 stride-2 transposed conv (82x98x70 geometry)
        else MM_LAUNCH(4, 4, 0, 0, 0, 0)
    } else { vg_set_error("vg_conv_mm: no kernel instance for %d classes x %d tiles per wave", d->nq, d->tpc); return VG_ERR_UNSUPPORTED; }
#undef MM_TPC
#undef MM_LAUNCH
#undef MM_LAUNCH_W
    return vg_check_launch("conv_mm");
}

namespace {
__global__ void __launch_bounds__(256)
gather_f32_k(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst, long long n) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int i = idx[e];
        dst[e] = i >= 0 ? src[i] : 0.f;
    }
}
}  // namespace

extern "C" int vg_gather_f32(const float* src, const int32_t* idx, float* dst, int64_t n, void* stream) {
    if (!src || !idx || !dst || n <= 0) { vg_set_error("vg_gather_f32: bad arguments"); return VG_ERR_ARG; }
    long long blocks = (n + 255) / 256; if (blocks > 1024) blocks = 1024;
    vg_launch(gather_f32_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (const int*)idx, dst, (long long)n);
    return vg_check_launch("gather_f32");
}
