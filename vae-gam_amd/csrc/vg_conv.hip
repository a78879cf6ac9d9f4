// vg_conv.hip -- direct 3-D convolution kernels for the VAE-GAM encoder/decoder (gfx950).
//
// Two kernel families cover every conv / transposed-conv forward and data gradient of
// vae_reg_GP.py:187-218 (see include/vaegam.h for the maths; weight gradients: vg_wgrad.hip):
//   corr3d      strided correlation (stride 1/2, leading zero padding)
//   tconv3d_s2  stride-2 transposed convolution in gather form (each thread owns a 2x2x2 output brick)
//
// Channel counts are 1/8/16, so one GEMM dimension is at most 16: these are register-tiled fp32
// VALU kernels.  A thread owns a small output brick x COT output channels (all accumulators in
// VGPRs); weights arrive through the scalar path (wave-uniform, pre-packed [ci][tap][co]); the
// input window of the brick is read straight into registers with the producer's ReLU / batch-norm
// affine applied on the way (activations are stored pre-activation), and neighbouring threads'
// overlapping windows are served by the CU's vector L1.
//
// Measured on MI355X (round 1, tools/layer_bench.py, convt5 forward at 128 x 8 x 39x47x33):
// the first version staged the haloed input tile through LDS once per channel chunk (two barriers
// per chunk, one wave per tile row): 785 us, of which 107 us was arithmetic -- the tile fill was
// latency-bound (4 waves per block, 152-byte rows, <20 KB in flight per CU).  Reading the window
// directly keeps 50-130 independent loads in flight per thread with no barrier and 2-4x the
// occupancy.  LDS staging is kept where a tile is re-read many times by different lanes
// (vg_wgrad.hip).
#include "vg_common.h"
#include <stdlib.h>
#include "../../include/vaegam.h"

namespace {

struct CorrParams {
    vg_conv_desc d;
    int TWG, TH, TD;            // threads of a block along w / h / d
    int tilesW, tilesH, tilesD;
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// ------------------------------------------------------------------------------------------
// corr3d
// ------------------------------------------------------------------------------------------
template <int COT, int KD, int KH, int KW, int S, int TDt, int THt, int TW, int COC>
__global__ void __launch_bounds__(256)
corr3d_direct_k(const float* __restrict__ x, const float* __restrict__ wpk, const float* __restrict__ bias,
         const float* __restrict__ in_scale, const float* __restrict__ in_shift,
         const float* __restrict__ mask_src, float* __restrict__ y, CorrParams p) {
    constexpr int KVOL = KD * KH * KW;
    constexpr int RW = (TW - 1) * S + KW;
    constexpr int RD = (TDt - 1) * S + KD;
    constexpr int RH = (THt - 1) * S + KH;
    const vg_conv_desc& d = p.d;
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    const int CO = COC ? COC : d.CO;             // all output channels (COC: compile-time); this block computes [co0, co0+COT)
    const int co0 = blockIdx.z * COT;
    int tile = blockIdx.x;
    const int twi = tile % p.tilesW; tile /= p.tilesW;
    const int thi = tile % p.tilesH; const int tdi = tile / p.tilesH;
    const int wg = tid % p.TWG; const int thl = (tid / p.TWG) % p.TH; const int tdl = tid / (p.TWG * p.TH);
    // first output of this thread's brick
    const int od_t = (tdi * p.TD + tdl) * TDt, oh_t = (thi * p.TH + thl) * THt, ow_t = (twi * p.TWG + wg) * TW;
    if (tdl >= p.TD || od_t >= d.OD || oh_t >= d.OH || ow_t >= d.OW) return;      // no barriers in this kernel
    const int id_t = od_t * S - d.pad_d, ih_t = oh_t * S - d.pad_h, iw_t = ow_t * S - d.pad_w;

    // window geometry is the same for every input channel: clamped row offsets / column indices and
    // their validity (zero padding and tile overhang) are computed once
    int roff[RD][RH]; bool rok[RD][RH];
#pragma unroll
    for (int dz = 0; dz < RD; ++dz)
#pragma unroll
        for (int hy = 0; hy < RH; ++hy) {
            const int id = id_t + dz, ih = ih_t + hy;
            rok[dz][hy] = id >= 0 && id < d.ID && ih >= 0 && ih < d.IH;
            roff[dz][hy] = (clampi(id, 0, d.ID - 1) * d.IH + clampi(ih, 0, d.IH - 1)) * d.IW;
        }
    int cof[RW]; bool cok[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) { const int iw = iw_t + i; cok[i] = iw >= 0 && iw < d.IW; cof[i] = clampi(iw, 0, d.IW - 1); }

    float acc[TDt][THt][TW][COT];
#pragma unroll
    for (int a = 0; a < TDt; ++a)
#pragma unroll
        for (int b = 0; b < THt; ++b)
#pragma unroll
            for (int j = 0; j < TW; ++j)
#pragma unroll
                for (int co = 0; co < COT; ++co) acc[a][b][j][co] = 0.f;

    const int g = (in_scale != nullptr) ? n / d.per_group : 0;
    const float lo = d.relu_in ? 0.f : -__builtin_inff();            // max(v, lo): ReLU or identity
    const size_t vol = (size_t)d.ID * d.IH * d.IW;
    // The window is consumed one d-slab (RH x RW values) at a time; the NEXT slab's loads are issued
    // before the current slab's arithmetic, so RH*RW independent loads are always in flight behind the
    // FMAs (software pipeline over the flattened (channel, dz) sequence, two register buffers).
    float buf[2][RH][RW];
    const float* __restrict__ xbase = x + (size_t)n * d.CI * vol;
#pragma unroll
    for (int hy = 0; hy < RH; ++hy)
#pragma unroll
        for (int i = 0; i < RW; ++i) buf[0][hy][i] = xbase[roff[0][hy] + cof[i]];
    for (int ci = 0; ci < d.CI; ++ci) {
        const float* __restrict__ xc = xbase + (size_t)ci * vol;
        const float* __restrict__ xn = xbase + (size_t)min(ci + 1, d.CI - 1) * vol;      // next channel (clamped)
        const float* __restrict__ wc = wpk + (size_t)ci * KVOL * CO + co0;
        float sc = 1.f, sh = 0.f;
        if (in_scale != nullptr) { sc = in_scale[g * d.CI + ci]; sh = in_shift[g * d.CI + ci]; }
#pragma unroll
        for (int dz = 0; dz < RD; ++dz) {
            // ---- prefetch the next slab
            {
                const float* src = (dz + 1 < RD) ? xc : xn;
                const int dzn = (dz + 1 < RD) ? dz + 1 : 0;
#pragma unroll
                for (int hy = 0; hy < RH; ++hy)
#pragma unroll
                    for (int i = 0; i < RW; ++i) buf[(dz + 1) & 1][hy][i] = src[roff[dzn][hy] + cof[i]];
            }
            // ---- consume the current slab
#pragma unroll
            for (int hy = 0; hy < RH; ++hy) {
                float seg[RW];
#pragma unroll
                for (int i = 0; i < RW; ++i) {
                    const float v = fmaf(vg_max(buf[dz & 1][hy][i], lo), sc, sh);
                    seg[i] = (rok[dz][hy] && cok[i]) ? v : 0.f;               // zero padding outside the input
                }
#pragma unroll
                for (int a = 0; a < TDt; ++a) {
                    const int kd = dz - a * S;
                    if (kd < 0 || kd >= KD) continue;
#pragma unroll
                    for (int b = 0; b < THt; ++b) {
                        const int kh = hy - b * S;
                        if (kh < 0 || kh >= KH) continue;
#pragma unroll
                        for (int kw = 0; kw < KW; ++kw) {
                            const int t = (kd * KH + kh) * KW + kw;
#pragma unroll
                            for (int co = 0; co < COT; ++co) {
                                const float wv = wc[t * CO + co];
#pragma unroll
                                for (int j = 0; j < TW; ++j)
                                    acc[a][b][j][co] = fmaf(seg[j * S + kw], wv, acc[a][b][j][co]);
                            }
                        }
                    }
                }
            }
        }
        if (RD & 1) {                              // odd slab count: the prefetched slab 0 of the next channel sits in buf[1]
#pragma unroll
            for (int hy = 0; hy < RH; ++hy)
#pragma unroll
                for (int i = 0; i < RW; ++i) buf[0][hy][i] = buf[1][hy][i];
        }
    }
    const size_t plane = (size_t)d.OH * d.OW;
    // fused ReLU backward of the producer: fetch all mask values first (clamped, unconditional loads that
    // stay in flight together), then apply + store
    if (mask_src) {
#pragma unroll
        for (int a = 0; a < TDt; ++a) {
            const int odc = min(od_t + a, d.OD - 1);
#pragma unroll
            for (int b = 0; b < THt; ++b) {
                const int ohc = min(oh_t + b, d.OH - 1);
                float m[COT][TW];
#pragma unroll
                for (int co = 0; co < COT; ++co) {
                    const size_t base = (((size_t)n * CO + co0 + co) * d.OD + odc) * plane + (size_t)ohc * d.OW;
#pragma unroll
                    for (int j = 0; j < TW; ++j) m[co][j] = mask_src[base + min(ow_t + j, d.OW - 1)];
                }
#pragma unroll
                for (int co = 0; co < COT; ++co)
#pragma unroll
                    for (int j = 0; j < TW; ++j)
                        if (!(m[co][j] > 0.f)) acc[a][b][j][co] = -__builtin_inff();      // marked, zeroed below
            }
        }
    }
#pragma unroll
    for (int a = 0; a < TDt; ++a) {
        const int od = od_t + a;
        if (od >= d.OD) continue;
#pragma unroll
        for (int b = 0; b < THt; ++b) {
            const int oh = oh_t + b;
            if (oh >= d.OH) continue;
#pragma unroll
            for (int co = 0; co < COT; ++co) {
                const float bv = bias ? bias[co0 + co] : 0.f;
                const size_t base = (((size_t)n * CO + co0 + co) * d.OD + od) * plane + (size_t)oh * d.OW;
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    const int ow = ow_t + j;
                    if (ow < d.OW) {
                        const float v = acc[a][b][j][co];
                        y[base + ow] = (v == -__builtin_inff()) ? 0.f : v + bv;
                    }
                }
            }
        }
    }
}

template <int COT, int KD, int KH, int KW, int S, int TDt, int THt, int TW, int COC = 0>
int launch_corr(const vg_conv_desc* d, const float* x, const float* wpk, const float* bias, const float* in_scale,
                const float* in_shift, const float* mask_src, float* y, hipStream_t s) {
    CorrParams p; p.d = *d;
    const int gw = vg_cdiv(d->OW, TW), gh = vg_cdiv(d->OH, THt), gd = vg_cdiv(d->OD, TDt);
    // a block covers a compact brick of outputs so that the overlapping input windows of its
    // threads hit the CU's L1: all of w (up to 16 thread columns), then h, then d
    p.TWG = gw < 16 ? gw : 16;
    int rem = 256 / p.TWG;
    p.TH = gh < rem ? gh : rem;
    if (p.TH > 8 && gd > 1) p.TH = 8;
    rem = 256 / (p.TWG * p.TH);
    p.TD = gd < rem ? gd : rem;
    if (p.TD < 1) p.TD = 1;
    p.tilesW = vg_cdiv(gw, p.TWG); p.tilesH = vg_cdiv(gh, p.TH); p.tilesD = vg_cdiv(gd, p.TD);
    if (d->CO % COT) { vg_set_error("corr3d: CO=%d not a multiple of %d", d->CO, COT); return VG_ERR_UNSUPPORTED; }
    const int threads = vg_cdiv(p.TWG * p.TH * p.TD, VG_WAVE) * VG_WAVE;
    dim3 grid(p.tilesW * p.tilesH * p.tilesD, d->N, d->CO / COT);
    vg_launch(corr3d_direct_k<COT, KD, KH, KW, S, TDt, THt, TW, COC>, grid, dim3(threads), 0, s,
              x, wpk, bias, in_scale, in_shift, mask_src, y, p);
    return vg_check_launch("corr3d_direct");
}

// ------------------------------------------------------------------------------------------
// corr3d, plane-staged.  For every layer of the 41x49x35 network one block can cover ALL of H and W
// of a few output planes, so the input a block needs per channel is LD whole input planes = ONE
// contiguous span of the tensor.  That span is copied flat into LDS by LDS-DMA (global_load_lds_dword:
// 256 contiguous bytes per wave-instruction, no VGPR round trip, every instruction of the chunk in
// flight at once, ~5 instructions of address arithmetic each) and keeps the tensor's own row/plane
// pitch; a thread then reads its window with immediate offsets, applying ReLU / batch-norm affine /
// zero-padding masks on the way out of LDS.
// (tools/micro/fill_bench.hip: this fill runs at 6-7 TB/s of tile bytes; a per-row gather of the same
// tile 3.6 TB/s; the first per-row version with run-time row decoding was scalar-ALU bound at 0.6 TB/s.)
// ------------------------------------------------------------------------------------------
struct CorrPlaneParams {
    vg_conv_desc d;
    int TWG, TH, TD;            // threads along w / h / d; the block covers all of OW and OH
    int tilesD;
    int LD;                     // input planes per chunk-channel
    int CCH;                    // channels per chunk
    int ch_floats;              // LDS floats per channel (LD*IH*IW rounded up)
    int nbuf;                   // 2: the next chunk's planes are DMA'd behind this chunk's FMAs (one barrier per chunk)
    // row slabs (planes too large for one block -- the 82x98x70 geometry): a block covers OHB output rows of its planes and stages
    // the LH input rows they touch, one contiguous span per plane at the LDS plane pitch lplane
    int OHB, nhb, LH, lplane;
};
constexpr int PLANE_SLACK = 8;  // floats in front of the LDS buffer: a window may start at iw = -pad

template <int COT, int KD, int KH, int KW, int S, int TDt, int THt, int TW, int COC, bool NOPRO = false>
__global__ void __launch_bounds__(256, (TDt * THt * TW > 16 ? 2 : 3))
corr3d_plane_k(const float* __restrict__ x, const float* __restrict__ wpk, const float* __restrict__ bias,
               const float* __restrict__ in_scale, const float* __restrict__ in_shift,
               const float* __restrict__ mask_src, float* __restrict__ y, CorrPlaneParams p) {
    VG_DYN_SMEM(float, lds_raw);
    float* lds = lds_raw + PLANE_SLACK;
    constexpr int KVOL = KD * KH * KW;
    constexpr int RW = (TW - 1) * S + KW;
    constexpr int RD = (TDt - 1) * S + KD;
    constexpr int RH = (THt - 1) * S + KH;
    const vg_conv_desc& d = p.d;
    const int tid = threadIdx.x;
    const int lane = tid % VG_WAVE, wave = vg_wave_id();
    const int n = blockIdx.y;
    const int CO = COC ? COC : d.CO;                 // compile-time channel count: per-tap weight offsets become load immediates
    const int co0 = blockIdx.z * COT;
    const int tdi = blockIdx.x / p.nhb, hb = blockIdx.x % p.nhb;
    const int wg = tid % p.TWG; const int thl = (tid / p.TWG) % p.TH; const int tdl = tid / (p.TWG * p.TH);
    const int od0 = tdi * p.TD * TDt, oh0 = hb * p.OHB;
    const int od_t = od0 + tdl * TDt, oh_t = oh0 + thl * THt, ow_t = wg * TW;
    const bool active = tdl < p.TD && od_t < d.OD && thl * THt < p.OHB && oh_t < d.OH && ow_t < d.OW;
    const int plane = d.IH * d.IW;
    // staged input rows of every plane: [r_lo, r_hi) = the rows the slab's windows touch, clipped to the tensor; a plane's LDS slot
    // starts with row r_lo.  Whole planes (one slab, tensor pitch): r_lo = 0, r_hi = IH, one span per channel.
    const int ih0 = oh0 * S - d.pad_h;
    const int r_lo = max(ih0, 0), r_hi = min(ih0 + p.LH, d.IH);
    const int lplane = p.lplane;
    const bool flat = (r_lo == 0 && r_hi == d.IH && lplane == plane);
    // input planes [ip0, ip0 + LD) clipped to the tensor; local plane l <-> input plane ip0 + l
    const int ip0 = od0 * S - d.pad_d;
    const int pl_lo = max(ip0, 0), pl_hi = min(ip0 + p.LD, d.ID);          // valid planes [pl_lo, pl_hi)
    const int npl = max(pl_hi - pl_lo, 0);
    const int nfl = flat ? npl * plane : max(r_hi - r_lo, 0) * d.IW;        // floats per span (flat: one span per channel, else one per plane)
    const int dst0 = (pl_lo - ip0) * lplane;                                // where the first staged plane lands inside the channel slot

    // this thread's window: LDS offsets of its rows (row index clamped, masked below) and validity
    // Validity of the window's rows / columns.  VEC_MASK: as VECTOR values -- a row factor (1 or 0) folded into the prologue's scale
    // and shift, a column bit mask (all ones or zero) and-ed onto the result.  As boolean predicates they are RD*RH*RW scalar
    // register PAIRS for the selects (96..192 of the 100 scalar registers), which the compiler parks in VGPR lanes and fetches back with
    // two v_readlane per staged element (ISA: 256-333 v_readlane per channel iteration beside 433 v_fmac / 720 v_pk_fma).  Measured on
    // MI355X (batch 64, 8 covariates): convt4's data gradient 657 -> 574 us, convt5's forward 696 -> 672 us; the 3x3x3 stride-1
    // 8-channel instance (convt3) ran 8-14 % SLOWER that way and keeps the predicates.
    constexpr bool VEC_MASK = (COT == 1) || (KD == 5);
    int roff[RD][RH]; float rokf[RD][RH]; unsigned cokm[RW]; bool rok[RD][RH]; bool cok[RW];
    const int iw_t = ow_t * S - d.pad_w;
#pragma unroll
    for (int dz = 0; dz < RD; ++dz)
#pragma unroll
        for (int hy = 0; hy < RH; ++hy) {
            const int id = od_t * S - d.pad_d + dz, ih = oh_t * S - d.pad_h + hy;
            rok[dz][hy] = id >= 0 && id < d.ID && ih >= 0 && ih < d.IH;
            rokf[dz][hy] = VEC_MASK ? vg_opaque(rok[dz][hy] ? 1.f : 0.f) : 0.f;
            // VEC_MASK multiplies instead of selecting: clamp to a plane that WAS staged (0 * real data is 0, 0 * LDS garbage may be NaN)
            const int lp = VEC_MASK ? clampi(id, pl_lo, max(pl_hi - 1, pl_lo)) - ip0 : clampi(id - ip0, 0, p.LD - 1);
            roff[dz][hy] = lp * lplane + (clampi(ih, r_lo, max(r_hi - 1, r_lo)) - r_lo) * d.IW + iw_t;
            // NOPRO (no ReLU / affine on the input: the data-gradient launches): a row outside the tensor reads the zeroed tail of the
            // channel slot instead, so an element costs ONE v_and (column mask) instead of max + fma + and
            // (applying the prologue in place in LDS after the copy and running the one-channel layer through this path was measured:
            // 216 fewer vector instructions per channel of ~850, but one more barrier-separated phase per channel: 555 -> 710 us)
            if (NOPRO && !rok[dz][hy]) roff[dz][hy] = p.ch_floats - 32;
        }
    if (NOPRO) {
        for (int c = 0; c < p.CCH * p.nbuf; ++c)
            for (int i = tid; i < 64; i += blockDim.x) lds[c * p.ch_floats + p.ch_floats - 64 + i] = 0.f;     // never touched by the DMA
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int iw = iw_t + i;
        cok[i] = iw >= 0 && iw < d.IW;
        cokm[i] = VEC_MASK ? vg_opaque(cok[i] ? 0xffffffffu : 0u) : 0u;
    }

    float acc[TDt][THt][TW][COT];
#pragma unroll
    for (int a = 0; a < TDt; ++a)
#pragma unroll
        for (int b = 0; b < THt; ++b)
#pragma unroll
            for (int j = 0; j < TW; ++j)
#pragma unroll
                for (int co = 0; co < COT; ++co) acc[a][b][j][co] = 0.f;

    const int g = (in_scale != nullptr) ? n / d.per_group : 0;
    const float lo = d.relu_in ? 0.f : -__builtin_inff();
    const size_t vol = (size_t)plane * d.ID;
    const float* __restrict__ xsrc = x + (size_t)n * d.CI * vol + (size_t)pl_lo * plane + (size_t)r_lo * d.IW;
    auto stage = [&](int c0, float* buf) {
#ifdef VG_ABLATE_DMA                                          // diagnostic builds (tools/diag/build_variant.sh): the compute phases alone
        return;
#endif
        const int cc = min(p.CCH, d.CI - c0);
        const int nw = (int)(blockDim.x / VG_WAVE);
        for (int c = 0; c < cc; ++c) {
            const float* src = xsrc + (size_t)(c0 + c) * vol;
            float* dst = buf + c * p.ch_floats + dst0;
            if (flat) vg_dma_block(src, dst, nfl, wave, nw, lane);
            else for (int pl = 0; pl < npl; ++pl) vg_dma_block(src + (size_t)pl * plane, dst + pl * lplane, nfl, wave, nw, lane);
        }
    };
    const int buf_floats = p.CCH * p.ch_floats;
    if (p.nbuf == 2) stage(0, lds);
    int kchunk = 0;
    for (int c0 = 0; c0 < d.CI; c0 += p.CCH, ++kchunk) {
        const int cc = min(p.CCH, d.CI - c0);
        const float* cur = lds;
        if (p.nbuf == 2) {
            // this chunk's DMAs (issued during the previous chunk's FMAs) have landed; every wave is past the previous
            // chunk, so the other buffer is free for the next one
            vg_dma_wait();
            __syncthreads();
            cur = lds + (kchunk & 1) * buf_floats;
            if (c0 + p.CCH < d.CI) stage(c0 + p.CCH, lds + ((kchunk + 1) & 1) * buf_floats);
        } else {
            __syncthreads();                              // previous chunk fully consumed
            stage(c0, lds);
            vg_dma_wait();
            __syncthreads();
        }
#ifdef VG_ABLATE_FMA                                          // diagnostic builds: the copy phases alone
        if (active && d.CI < 0) {
#else
        if (active) {
#endif
            for (int c = 0; c < cc; ++c) {
                const int ci = c0 + c;
                const float* __restrict__ wc = wpk + (size_t)ci * KVOL * CO + co0;
                float sc = 1.f, sh = 0.f;
                if (in_scale != nullptr) { sc = in_scale[g * d.CI + ci]; sh = in_shift[g * d.CI + ci]; }
                const float* tl = cur + c * p.ch_floats;
#pragma unroll
                for (int dz = 0; dz < RD; ++dz) {
#pragma unroll
                    for (int hy = 0; hy < RH; ++hy) {
                        const float* row = tl + roff[dz][hy];
                        float seg[RW];
                        if constexpr (NOPRO) {
#pragma unroll
                            for (int i = 0; i < RW; ++i) seg[i] = vg_and(row[i], cokm[i]);
                        } else if constexpr (VEC_MASK) {
                            const float scr = sc * rokf[dz][hy], shr = sh * rokf[dz][hy];      // (clamped rows hold real, finite data)
#pragma unroll
                            for (int i = 0; i < RW; ++i) seg[i] = vg_and(fmaf(vg_max(row[i], lo), scr, shr), cokm[i]);
                        } else {
#pragma unroll
                            for (int i = 0; i < RW; ++i) {
                                const float v = fmaf(vg_max(row[i], lo), sc, sh);
                                seg[i] = (rok[dz][hy] && cok[i]) ? v : 0.f;
                            }
                        }
#pragma unroll
                        for (int a = 0; a < TDt; ++a) {
                            const int kd = dz - a * S;
                            if (kd < 0 || kd >= KD) continue;
#pragma unroll
                            for (int b = 0; b < THt; ++b) {
                                const int kh = hy - b * S;
                                if (kh < 0 || kh >= KH) continue;
#pragma unroll
                                for (int kw = 0; kw < KW; ++kw) {
                                    const int t = (kd * KH + kh) * KW + kw;
#pragma unroll
                                    for (int co = 0; co < COT; ++co) {
                                        const float wv = wc[t * CO + co];
#pragma unroll
                                        for (int j = 0; j < TW; ++j)
                                            acc[a][b][j][co] = fmaf(seg[j * S + kw], wv, acc[a][b][j][co]);
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    if (!active) return;
    const size_t oplane = (size_t)d.OH * d.OW;
    if (mask_src) {
#pragma unroll
        for (int a = 0; a < TDt; ++a) {
            const int odc = min(od_t + a, d.OD - 1);
#pragma unroll
            for (int b = 0; b < THt; ++b) {
                const int ohc = min(oh_t + b, d.OH - 1);
                float m[COT][TW];
#pragma unroll
                for (int co = 0; co < COT; ++co) {
                    const size_t base = (((size_t)n * CO + co0 + co) * d.OD + odc) * oplane + (size_t)ohc * d.OW;
#pragma unroll
                    for (int j = 0; j < TW; ++j) m[co][j] = mask_src[base + min(ow_t + j, d.OW - 1)];
                }
#pragma unroll
                for (int co = 0; co < COT; ++co)
#pragma unroll
                    for (int j = 0; j < TW; ++j)
                        if (!(m[co][j] > 0.f)) acc[a][b][j][co] = -__builtin_inff();
            }
        }
    }
#pragma unroll
    for (int a = 0; a < TDt; ++a) {
        const int od = od_t + a;
        if (od >= d.OD) continue;
#pragma unroll
        for (int b = 0; b < THt; ++b) {
            const int oh = oh_t + b;
            if (oh >= d.OH) continue;
#pragma unroll
            for (int co = 0; co < COT; ++co) {
                const float bv = bias ? bias[co0 + co] : 0.f;
                const size_t base = (((size_t)n * CO + co0 + co) * d.OD + od) * oplane + (size_t)oh * d.OW;
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    const int ow = ow_t + j;
                    if (ow < d.OW) {
                        const float v = acc[a][b][j][co];
                        y[base + ow] = (v == -__builtin_inff()) ? 0.f : v + bv;
                    }
                }
            }
        }
    }
}

// plane-staged launch; returns -1 if the geometry does not fit (caller falls back to the direct kernel)
template <int COT, int KD, int KH, int KW, int S, int TDt, int THt, int TW, int COC = 0>
int launch_corr_plane(const vg_conv_desc* d, const float* x, const float* wpk, const float* bias, const float* in_scale,
                      const float* in_shift, const float* mask_src, float* y, hipStream_t s) {
    CorrPlaneParams p; p.d = *d;
    const int gw = vg_cdiv(d->OW, TW), gd = vg_cdiv(d->OD, TDt);
    if (gw > 256) return -1;
    const int plane = d->IH * d->IW;
    const size_t budget = 48 * 1024;
    // rows: one block spans all of H where the threads and ONE channel's planes allow it (every layer of the 41x49x35 network);
    // otherwise the fewest row slabs that do (82x98x70)
    int nhb = 1, ohb = d->OH, gh = vg_cdiv(d->OH, THt), lh = d->IH, lplane = plane;
    for (;; ++nhb) {
        ohb = vg_cdiv(vg_cdiv(d->OH, nhb), THt) * THt;
        gh = ohb / THt;
        if (nhb > 1) { lh = (ohb - 1) * S + KH; if (lh > d->IH) lh = d->IH; lplane = ((lh * d->IW + 3) / 4) * 4; }
        const size_t need = ((size_t)((TDt - 1) * S + KD) * lplane + 64 + 64) * sizeof(float);     // one tile of output planes deep, one channel
        if (gw * gh <= 256 && need + (PLANE_SLACK + 64) * sizeof(float) <= budget) break;
        if (ohb <= THt) return -1;
    }
    nhb = vg_cdiv(d->OH, ohb);
    p.TWG = gw; p.TH = gh; p.OHB = ohb; p.nhb = nhb; p.LH = (nhb == 1) ? d->IH + d->pad_h : lh; p.lplane = lplane;
    int td = 256 / (gw * gh);
    if (td > gd) td = gd;
    for (; td >= 1; --td) {
        const int LD = (td * TDt - 1) * S + KD;
        if (((size_t)LD * lplane + 64) * sizeof(float) + 64 <= budget) break;
    }
    if (td < 1) return -1;
    p.TD = td; p.LD = (td * TDt - 1) * S + KD;
    p.tilesD = vg_cdiv(gd, p.TD);
    p.ch_floats = ((p.LD * lplane + 63) / 64) * 64 + 64;  // slot per channel: whole 256-byte DMA groups + slack for window over-reads
    // Two buffers (the next chunk in flight behind this chunk's FMAs, one barrier per chunk) only for the large-tile instances.
    // Measured on MI355X: no gain where the chunk is small (convt3/convt4: 3 blocks per CU already hide the fill) and a
    // loss where it doubles a 25 KB slot (convt5 forward 244 -> 341 us: only two 52 KB blocks fit a CU).
    const size_t budget2 = (TDt * THt * TW > 16 ? 78 : 52) * 1024;      // two blocks per CU there (registers), three otherwise
    p.nbuf = (TDt * THt * TW > 16 && d->CI > 1 && 2 * (size_t)p.ch_floats * sizeof(float) + (PLANE_SLACK + 64) * sizeof(float) <= budget2) ? 2 : 1;
    const size_t bud = p.nbuf == 2 ? budget2 : budget;
    int cch = (int)((bud - (PLANE_SLACK + 64) * sizeof(float)) / ((size_t)p.ch_floats * sizeof(float) * p.nbuf));
    if (cch < 1) cch = 1;
    if (cch > d->CI) cch = d->CI;
    if (p.nbuf == 2 && cch >= d->CI) { p.nbuf = 1; }      // everything fits one chunk: nothing to overlap
    p.CCH = cch;
    // (Round 3, measured and NOT kept: split-phase staging for the layers whose chunk is one channel in one buffer -- planes [0, h) of the
    // next channel copied behind the FMAs on planes [h, LD) of this one and the other way round, copies issued from inline assembly so
    // that the compiler does not wait for them at the next LDS read.  With the copy / FMA phases ablated in turn convt4's data gradient is
    // 275 us of copies + 323 us of FMAs in 454 us and convt5 forward 269 + 443 in 535, so perfect overlap would save ~0.2 ms; the split
    // gave 494 and 553-583 us: half a channel of FMAs (~1 us) is shorter than a copy's latency, and where a block spans two output-plane
    // groups its waves need different halves, so each phase waits for its slowest wave.  What would work is a second buffer, which does
    // not fit beside the co-resident blocks.)
    const size_t shmem = (size_t)p.ch_floats * cch * p.nbuf * sizeof(float) + (PLANE_SLACK + 64) * sizeof(float);
    if (d->CO % COT) { vg_set_error("corr3d: CO=%d not a multiple of %d", d->CO, COT); return VG_ERR_UNSUPPORTED; }
    const int threads = vg_cdiv(p.TWG * p.TH * p.TD, VG_WAVE) * VG_WAVE;
    dim3 grid(p.tilesD * p.nhb, d->N, d->CO / COT);
    if constexpr (KD == 5) {                                     // (the instances built with vector masks; the one-channel one always has a prologue)
        if (in_scale == nullptr && !d->relu_in) {
            vg_launch(corr3d_plane_k<COT, KD, KH, KW, S, TDt, THt, TW, COC, true>, grid, dim3(threads), shmem, s,
                      x, wpk, bias, in_scale, in_shift, mask_src, y, p);
            return vg_check_launch("corr3d_plane");
        }
    }
    vg_launch(corr3d_plane_k<COT, KD, KH, KW, S, TDt, THt, TW, COC>, grid, dim3(threads), shmem, s,
              x, wpk, bias, in_scale, in_shift, mask_src, y, p);
    return vg_check_launch("corr3d_plane");
}

// ------------------------------------------------------------------------------------------
// tconv3d_s2: thread (jd,jh,jw) owns outputs q = 2j + r (r in {0,1}^3), o = q - pad.
// tap k = r + 2m  <->  input i = j - m.
// ------------------------------------------------------------------------------------------
struct TconvParams {
    vg_conv_desc d;
    int TJW, TJH, TJD;
    int tilesW, tilesH, tilesD;
    int JD, JH, JW;             // number of j positions per dim
};

template <int COT, int KD, int KH, int KW, int COC>
__global__ void __launch_bounds__(256, 4)
tconv3d_s2_k(const float* __restrict__ x, const float* __restrict__ wpk, const float* __restrict__ bias,
             const float* __restrict__ in_scale, const float* __restrict__ in_shift,
             const float* __restrict__ mask_src, float* __restrict__ y, TconvParams p,
             double* __restrict__ stats_part, int stats_relu, int stats_pg) {
    constexpr int KVOL = KD * KH * KW;
    constexpr int MD = (KD + 1) / 2, MH = (KH + 1) / 2, MW = (KW + 1) / 2;
    const vg_conv_desc& d = p.d;
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    int tile = blockIdx.x;
    const int twi = tile % p.tilesW; tile /= p.tilesW;
    const int thi = tile % p.tilesH; const int tdi = tile / p.tilesH;
    // COC: the layer's channel count as a compile-time constant (0 = run-time): the per-tap weight offsets t*CO then fold into the
    // scalar loads' immediates -- with a run-time CO every one of the KVOL*CI weight loads pays 64-bit scalar address arithmetic
    // (rocprofv3: 1855 SALU vs 930 v_pk_fma per wavefront on convt4)
    const int CO = COC ? COC : d.CO;
    const int co0 = blockIdx.z * COT;
    const int jwl = tid % p.TJW; const int jhl = (tid / p.TJW) % p.TJH; const int jdl = tid / (p.TJW * p.TJH);
    const int jd = tdi * p.TJD + jdl, jh = thi * p.TJH + jhl, jw = twi * p.TJW + jwl;
    // threads past the tile / the volume stay alive (clamped loads, no stores): the optional statistics epilogue reduces
    // over whole wavefronts
    const bool active = !(jdl >= p.TJD || jd >= p.JD || jh >= p.JH || jw >= p.JW);
    if (!active && stats_part == nullptr) return;                                  // no barriers in this kernel

    // input i = j - m per dim: clamped offsets + validity, shared by all channels
    int roff[MD][MH]; bool rok[MD][MH];
#pragma unroll
    for (int md = 0; md < MD; ++md)
#pragma unroll
        for (int mh = 0; mh < MH; ++mh) {
            const int id = jd - md, ih = jh - mh;
            rok[md][mh] = id >= 0 && id < d.ID && ih >= 0 && ih < d.IH;
            roff[md][mh] = (clampi(id, 0, d.ID - 1) * d.IH + clampi(ih, 0, d.IH - 1)) * d.IW;
        }
    int cof[MW]; bool cok[MW];
#pragma unroll
    for (int mw = 0; mw < MW; ++mw) { const int iw = jw - mw; cok[mw] = iw >= 0 && iw < d.IW; cof[mw] = clampi(iw, 0, d.IW - 1); }

    float acc[2][2][2][COT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int co = 0; co < COT; ++co) acc[a][b][c][co] = 0.f;

    const int g = (in_scale != nullptr) ? n / d.per_group : 0;
    const float lo = d.relu_in ? 0.f : -__builtin_inff();
    const size_t vol = (size_t)d.ID * d.IH * d.IW;
    // software pipeline over channels: the next channel's window is in flight behind this channel's FMAs
    // (the weights are wave-uniform scalar loads used once per lane, requested right in front of their FMAs -- nine
    // `s_waitcnt lgkmcnt(0)` per channel in the ISA.  Requesting tap group g+1's weights before group g's FMAs was measured:
    // convt2 forward 245 -> 261 us; the four waves per SIMD already cover that latency.)
    float nxt[MD][MH][MW];
    const float* __restrict__ xbase = x + (size_t)n * d.CI * vol;
#pragma unroll
    for (int md = 0; md < MD; ++md)
#pragma unroll
        for (int mh = 0; mh < MH; ++mh)
#pragma unroll
            for (int mw = 0; mw < MW; ++mw) nxt[md][mh][mw] = xbase[roff[md][mh] + cof[mw]];
    for (int ci = 0; ci < d.CI; ++ci) {
        const float* __restrict__ xn = xbase + (size_t)min(ci + 1, d.CI - 1) * vol;
        const float* __restrict__ wc = wpk + (size_t)ci * KVOL * CO + co0;
        float sc = 1.f, sh = 0.f;
        if (in_scale != nullptr) { sc = in_scale[g * d.CI + ci]; sh = in_shift[g * d.CI + ci]; }
        float xin[MD][MH][MW];
#pragma unroll
        for (int md = 0; md < MD; ++md)
#pragma unroll
            for (int mh = 0; mh < MH; ++mh)
#pragma unroll
                for (int mw = 0; mw < MW; ++mw) {
                    const float v = fmaf(vg_max(nxt[md][mh][mw], lo), sc, sh);
                    xin[md][mh][mw] = (rok[md][mh] && cok[mw]) ? v : 0.f;
                    nxt[md][mh][mw] = xn[roff[md][mh] + cof[mw]];
                }
#pragma unroll
        for (int md = 0; md < MD; ++md)
#pragma unroll
            for (int mh = 0; mh < MH; ++mh)
#pragma unroll
                for (int mw = 0; mw < MW; ++mw) {
                    const float xv = xin[md][mh][mw];
#pragma unroll
                    for (int rd = 0; rd < 2; ++rd) {
                        const int kd = rd + 2 * md;
                        if (kd >= KD) continue;
#pragma unroll
                        for (int rh = 0; rh < 2; ++rh) {
                            const int kh = rh + 2 * mh;
                            if (kh >= KH) continue;
#pragma unroll
                            for (int rw = 0; rw < 2; ++rw) {
                                const int kw = rw + 2 * mw;
                                if (kw >= KW) continue;
                                const int t = (kd * KH + kh) * KW + kw;
#pragma unroll
                                for (int co = 0; co < COT; ++co)
                                    acc[rd][rh][rw][co] = fmaf(xv, wc[t * CO + co], acc[rd][rh][rw][co]);
                            }
                        }
                    }
                }
    }
    const size_t plane = (size_t)d.OH * d.OW;
    if (mask_src) {                                  // batched, clamped mask fetch (see corr3d_k)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int odc = min(max(2 * jd + rd - d.pad_d, 0), d.OD - 1);
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                const int ohc = min(max(2 * jh + rh - d.pad_h, 0), d.OH - 1);
                float m[COT][2];
#pragma unroll
                for (int co = 0; co < COT; ++co) {
                    const size_t base = (((size_t)n * CO + co0 + co) * d.OD + odc) * plane + (size_t)ohc * d.OW;
#pragma unroll
                    for (int rw = 0; rw < 2; ++rw) m[co][rw] = mask_src[base + min(max(2 * jw + rw - d.pad_w, 0), d.OW - 1)];
                }
#pragma unroll
                for (int co = 0; co < COT; ++co)
#pragma unroll
                    for (int rw = 0; rw < 2; ++rw)
                        if (!(m[co][rw] > 0.f)) acc[rd][rh][rw][co] = -__builtin_inff();
            }
        }
    }
    float st_s[COT], st_q[COT];                       // optional: sum / sum of squares of relu?(y) per channel (next layer's BN)
#pragma unroll
    for (int co = 0; co < COT; ++co) { st_s[co] = 0.f; st_q[co] = 0.f; }
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int od = 2 * jd + rd - d.pad_d;
        if (!active || od < 0 || od >= d.OD) continue;
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
            const int oh = 2 * jh + rh - d.pad_h;
            if (oh < 0 || oh >= d.OH) continue;
#pragma unroll
            for (int co = 0; co < COT; ++co) {
                const float bv = bias ? bias[co0 + co] : 0.f;
                const size_t base = (((size_t)n * CO + co0 + co) * d.OD + od) * plane + (size_t)oh * d.OW;
                // the thread's two outputs along w are adjacent: one 8-byte store per (row, channel) where both exist
                const int ow0 = 2 * jw - d.pad_w;
                float o2[2];
#pragma unroll
                for (int rw = 0; rw < 2; ++rw) {
                    const float v = acc[rd][rh][rw][co];
                    o2[rw] = (v == -__builtin_inff()) ? 0.f : v + bv;
                    if (ow0 + rw >= 0 && ow0 + rw < d.OW) {
                        const float h = stats_relu ? vg_max(o2[rw], 0.f) : o2[rw];
                        st_s[co] += h; st_q[co] = fmaf(h, h, st_q[co]);
                    }
                }
                if (ow0 >= 0 && ow0 + 1 < d.OW) vg_store2(y + base + ow0, o2[0], o2[1]);
                else {
                    if (ow0 >= 0 && ow0 < d.OW) y[base + ow0] = o2[0];
                    if (ow0 + 1 >= 0 && ow0 + 1 < d.OW) y[base + ow0 + 1] = o2[1];
                }
            }
        }
    }
    if (stats_part) {
        // one partial per WAVEFRONT (no barrier): part[((g*CO + c)*chunks + chunk)*2 + {0,1}], chunk = ((n % pg)*tiles + tile)*nw + wave
        const int lane = tid % VG_WAVE, wv = tid / VG_WAVE, nw = blockDim.x / VG_WAVE;
        const int g = n / stats_pg;
        const size_t chunks = (size_t)stats_pg * gridDim.x * nw;
        const size_t chunk = ((size_t)(n % stats_pg) * gridDim.x + blockIdx.x) * nw + wv;
#pragma unroll
        for (int co = 0; co < COT; ++co) {
            float a = st_s[co], b = st_q[co];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
            if (lane == 0) {
                double* dst = stats_part + (((size_t)g * CO + co0 + co) * chunks + chunk) * 2;
                dst[0] = (double)a; dst[1] = (double)b;
            }
        }
    }
}

template <int COT, int KD, int KH, int KW, int COC = 0>
int launch_tconv(const vg_conv_desc* d, const float* x, const float* wpk, const float* bias, const float* in_scale,
                 const float* in_shift, const float* mask_src, float* y, hipStream_t s,
                 double* stats_part = nullptr, int stats_relu = 0, int stats_pg = 1, int64_t* chunks_only = nullptr) {
    TconvParams p; p.d = *d;
    p.JD = (d->OD + d->pad_d + 1) / 2; p.JH = (d->OH + d->pad_h + 1) / 2; p.JW = (d->OW + d->pad_w + 1) / 2;
    p.TJW = p.JW < 32 ? p.JW : 32;
    int rem = 256 / p.TJW;
    p.TJH = p.JH < rem ? p.JH : rem;
    rem = 256 / (p.TJW * p.TJH);
    p.TJD = p.JD < rem ? p.JD : rem;
    if (p.TJD < 1) p.TJD = 1;
    p.tilesW = vg_cdiv(p.JW, p.TJW); p.tilesH = vg_cdiv(p.JH, p.TJH); p.tilesD = vg_cdiv(p.JD, p.TJD);
    if (d->CO % COT) { vg_set_error("tconv3d_s2: CO=%d not a multiple of %d", d->CO, COT); return VG_ERR_UNSUPPORTED; }
    const int threads = vg_cdiv(p.TJW * p.TJH * p.TJD, VG_WAVE) * VG_WAVE;
    dim3 grid(p.tilesW * p.tilesH * p.tilesD, d->N, d->CO / COT);
    if (chunks_only) { *chunks_only = (int64_t)stats_pg * grid.x * (threads / VG_WAVE); return VG_OK; }
    vg_launch(tconv3d_s2_k<COT, KD, KH, KW, COC>, grid, dim3(threads), 0, s,
              x, wpk, bias, in_scale, in_shift, mask_src, y, p, stats_part, stats_relu, stats_pg);
    return vg_check_launch("tconv3d_s2");
}

}  // namespace

static int check_desc(const vg_conv_desc* d, const void* x, const void* w, const void* y, const char* who) {
    if (!d || !x || !w || !y) { vg_set_error("%s: null argument", who); return VG_ERR_ARG; }
    if (d->N <= 0 || d->CI <= 0 || d->CO <= 0 || d->ID <= 0 || d->IH <= 0 || d->IW <= 0 || d->OD <= 0 || d->OH <= 0 ||
        d->OW <= 0 || (d->stride != 1 && d->stride != 2)) {
        vg_set_error("%s: bad shape N=%d CI=%d CO=%d in=%dx%dx%d out=%dx%dx%d stride=%d", who, d->N, d->CI, d->CO,
                     d->ID, d->IH, d->IW, d->OD, d->OH, d->OW, d->stride);
        return VG_ERR_ARG;
    }
    if (d->N > 65535) { vg_set_error("%s: N=%d exceeds grid.y", who, d->N); return VG_ERR_ARG; }
    return VG_OK;
}

extern "C" int vg_corr3d(const vg_conv_desc* d, const float* x, const float* wpk, const float* bias,
                         const float* in_scale, const float* in_shift, const float* mask_src, float* y, void* stream) {
    int rc = check_desc(d, x, wpk, y, "vg_corr3d");
    if (rc) return rc;
    if ((in_scale == nullptr) != (in_shift == nullptr) || (in_scale && d->per_group <= 0)) {
        vg_set_error("vg_corr3d: in_scale/in_shift/per_group inconsistent"); return VG_ERR_ARG;
    }
    // every output position must only read inside [ -pad, I ) -- the tile stager zero-fills the rest,
    // but an output size that does not match the geometry is a caller bug worth reporting.
    const int need_d = (d->OD - 1) * d->stride + d->KD - d->pad_d, need_h = (d->OH - 1) * d->stride + d->KH - d->pad_h,
              need_w = (d->OW - 1) * d->stride + d->KW - d->pad_w;
    if (need_d > d->ID + d->KD || need_h > d->IH + d->KH || need_w > d->IW + d->KW) {
        vg_set_error("vg_corr3d: output %dx%dx%d inconsistent with input %dx%dx%d", d->OD, d->OH, d->OW, d->ID, d->IH, d->IW);
        return VG_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    // "small" launches (few output positions) trade register tiling for parallelism: fewer outputs and
    // fewer output channels per thread, the channel groups spread over grid.z
    const long long pos = (long long)d->N * d->OD * d->OH * d->OW;
    const bool small = pos * d->CO < (long long)1536 * 1024;
    const bool k333 = d->KD == 3 && d->KH == 3 && d->KW == 3;
#define CORR(COT, KD, KH, KW, S, TDt, THt, TW) \
    return launch_corr<COT, KD, KH, KW, S, TDt, THt, TW>(d, x, wpk, bias, in_scale, in_shift, mask_src, y, s)
#define CORR_LDS(COT, KD, KH, KW, S, TDt, THt, TW) \
    { int r_ = d->CO == COT ? launch_corr_plane<COT, KD, KH, KW, S, TDt, THt, TW, COT>(d, x, wpk, bias, in_scale, in_shift, mask_src, y, s) \
             : d->CO == 2 * COT ? launch_corr_plane<COT, KD, KH, KW, S, TDt, THt, TW, 2 * COT>(d, x, wpk, bias, in_scale, in_shift, mask_src, y, s) \
             : launch_corr_plane<COT, KD, KH, KW, S, TDt, THt, TW, 0>(d, x, wpk, bias, in_scale, in_shift, mask_src, y, s); \
      if (r_ >= 0) return r_; \
      return launch_corr<COT, KD, KH, KW, S, TDt, THt, TW>(d, x, wpk, bias, in_scale, in_shift, mask_src, y, s); }
#define CORR_LDS_RT(COT, KD, KH, KW, S, TDt, THt, TW) /* run-time channel count: measured faster for the 3x3x3 stride-1 8-wide instance */ \
    { int r_ = launch_corr_plane<COT, KD, KH, KW, S, TDt, THt, TW, 0>(d, x, wpk, bias, in_scale, in_shift, mask_src, y, s); \
      if (r_ >= 0) return r_; \
      if (d->CO == COT) return launch_corr<COT, KD, KH, KW, S, TDt, THt, TW, COT>(d, x, wpk, bias, in_scale, in_shift, mask_src, y, s); \
      return launch_corr<COT, KD, KH, KW, S, TDt, THt, TW>(d, x, wpk, bias, in_scale, in_shift, mask_src, y, s); }
    if (k333 && d->stride == 1) {
        // one output channel (the decoder's last layer): 4x2x4 outputs per thread, 6 input planes per 4 output planes (input fetched
        // 1.5x instead of 2x), two blocks per CU with the next channel's planes in flight: 803 -> 735 us at batch 64 / 8 covariates
        if (d->CO == 1) CORR_LDS(1, 3, 3, 3, 1, 4, 2, 4);
        if (d->CO % 8 == 0 && !small) CORR_LDS_RT(8, 3, 3, 3, 1, 1, 1, 4);
        if (d->CO % 4 == 0 && small) CORR(4, 3, 3, 3, 1, 1, 1, 2);
    }
    if (k333 && d->stride == 2) {
        if (d->CO % 8 == 0 && !small) CORR_LDS(8, 3, 3, 3, 2, 1, 1, 4);
        if (d->CO % 4 == 0 && small) CORR(4, 3, 3, 3, 2, 1, 1, 2);
    }
    if (d->KD == 5 && d->KH == 3 && d->KW == 3 && d->stride == 2 && d->CO % 8 == 0) CORR_LDS(8, 5, 3, 3, 2, 1, 1, 4);
    if (d->KD == 4 && d->KH == 4 && d->KW == 4 && d->stride == 2 && d->CO % 8 == 0) CORR_LDS(8, 4, 4, 4, 2, 1, 1, 4);
#undef CORR
#undef CORR_LDS
    vg_set_error("vg_corr3d: no kernel instance for CO=%d k=%dx%dx%d stride=%d", d->CO, d->KD, d->KH, d->KW, d->stride);
    return VG_ERR_UNSUPPORTED;
}

static int tconv_dispatch(const vg_conv_desc* d, const float* x, const float* wpk, const float* bias, const float* in_scale,
                          const float* in_shift, const float* mask_src, float* y, hipStream_t s,
                          double* stats_part, int stats_relu, int stats_pg, int64_t* chunks_only) {
    const long long pos = (long long)d->N * d->OD * d->OH * d->OW;
    const bool small = pos * d->CO < (long long)8 * 1024 * 1024;
#define TCONV(COT, KD, KH, KW) \
    { if (d->CO == 8) return launch_tconv<COT, KD, KH, KW, 8>(d, x, wpk, bias, in_scale, in_shift, mask_src, y, s, stats_part, stats_relu, stats_pg, chunks_only); \
      if (d->CO == 16) return launch_tconv<COT, KD, KH, KW, 16>(d, x, wpk, bias, in_scale, in_shift, mask_src, y, s, stats_part, stats_relu, stats_pg, chunks_only); \
      return launch_tconv<COT, KD, KH, KW, 0>(d, x, wpk, bias, in_scale, in_shift, mask_src, y, s, stats_part, stats_relu, stats_pg, chunks_only); }
    if (d->KD == 3 && d->KH == 3 && d->KW == 3) {
        if (d->CO % 8 == 0 && !small) TCONV(8, 3, 3, 3);
        if (d->CO % 4 == 0 && small) TCONV(4, 3, 3, 3);
    }
    if (d->KD == 5 && d->KH == 3 && d->KW == 3 && d->CO % 8 == 0) TCONV(8, 5, 3, 3);
    if (d->KD == 4 && d->KH == 4 && d->KW == 4 && d->CO % 8 == 0) TCONV(8, 4, 4, 4);
#undef TCONV
    vg_set_error("vg_tconv3d_s2: no kernel instance for CO=%d k=%dx%dx%d", d->CO, d->KD, d->KH, d->KW);
    return VG_ERR_UNSUPPORTED;
}

static int tconv_check(const vg_conv_desc* d, const float* x, const float* wpk, const float* in_scale, const float* in_shift,
                       const float* y) {
    int rc = check_desc(d, x, wpk, y, "vg_tconv3d_s2");
    if (rc) return rc;
    if (d->stride != 2) { vg_set_error("vg_tconv3d_s2: stride must be 2"); return VG_ERR_ARG; }
    if ((in_scale == nullptr) != (in_shift == nullptr) || (in_scale && d->per_group <= 0)) {
        vg_set_error("vg_tconv3d_s2: in_scale/in_shift/per_group inconsistent"); return VG_ERR_ARG;
    }
    return VG_OK;
}

extern "C" int vg_tconv3d_s2(const vg_conv_desc* d, const float* x, const float* wpk, const float* bias,
                             const float* in_scale, const float* in_shift, const float* mask_src, float* y, void* stream) {
    int rc = tconv_check(d, x, wpk, in_scale, in_shift, y);
    if (rc) return rc;
    return tconv_dispatch(d, x, wpk, bias, in_scale, in_shift, mask_src, y, (hipStream_t)stream, nullptr, 0, 1, nullptr);
}

extern "C" int64_t vg_tconv3d_s2_stats_chunks(const vg_conv_desc* d, int32_t stats_per_group) {
    if (!d || stats_per_group <= 0 || d->N % stats_per_group) return -1;
    int64_t chunks = 0;
    float dummy;
    int rc = tconv_dispatch(d, &dummy, &dummy, nullptr, nullptr, nullptr, nullptr, &dummy, nullptr, nullptr, 0, stats_per_group, &chunks);
    return rc ? -1 : chunks;
}

extern "C" int vg_tconv3d_s2_stats(const vg_conv_desc* d, const float* x, const float* wpk, const float* bias,
                                   const float* in_scale, const float* in_shift, float* y, int32_t stats_per_group,
                                   int32_t stats_relu, double* stats_part, void* stream) {
    int rc = tconv_check(d, x, wpk, in_scale, in_shift, y);
    if (rc) return rc;
    if (!stats_part || stats_per_group <= 0 || d->N % stats_per_group) { vg_set_error("vg_tconv3d_s2_stats: bad statistics arguments"); return VG_ERR_ARG; }
    return tconv_dispatch(d, x, wpk, bias, in_scale, in_shift, nullptr, y, (hipStream_t)stream, stats_part, stats_relu,
                          stats_per_group, nullptr);
}
