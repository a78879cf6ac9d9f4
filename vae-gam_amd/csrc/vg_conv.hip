// vg_conv.hip -- direct 3-D convolution kernels for the VAE-GAM encoder/decoder (gfx950).
//
// Three kernel families cover every conv / transposed-conv forward, data-gradient and
// weight-gradient of vae_reg_GP.py:187-218 (see include/vaegam.h for the maths):
//   corr3d      strided correlation (stride 1/2, leading zero padding)
//   tconv3d_s2  stride-2 transposed convolution in gather form (each thread owns a 2x2x2 output brick)
//   wgrad3d     weight gradient, persistent blocks + deterministic second-stage reduction
//
// Channel counts are 1/8/16, so one GEMM dimension is at most 16: the kernels are register-tiled
// fp32 VALU kernels (each thread holds all CO accumulators of a small output brick), weights come
// in through the scalar path (wave-uniform, pre-packed [ci][tap][co]) and the input tile with its
// halo is staged once per channel chunk in LDS, with the producer's ReLU / batch-norm affine
// applied while staging (activations are stored pre-activation).
#include "vg_common.h"
#include "../../include/vaegam.h"

namespace {

struct CorrParams {
    vg_conv_desc d;
    int TWG, TH, TD;            // threads of a block along w / h / d
    int tilesW, tilesH, tilesD;
    int LD, LH, LW, LWp;        // LDS tile (per channel)
    int CCH;                    // channels staged per chunk
};

// ------------------------------------------------------------------------------------------
// staging: one wave per LDS row, lanes along w.  Applies prologue; zero outside the input.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_tile(float* lds, const float* __restrict__ x,
                                           const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                           const vg_conv_desc& d, int n, int c0, int cc,
                                           int id0, int ih0, int iw0, int LD, int LH, int LW, int LWp) {
    const int lane = threadIdx.x % VG_WAVE, wave = threadIdx.x / VG_WAVE, nwaves = blockDim.x / VG_WAVE;
    const int rows = cc * LD * LH;
    const int g = (in_scale != nullptr) ? n / d.per_group : 0;
    for (int r = wave; r < rows; r += nwaves) {
        const int hy = r % LH; const int t = r / LH; const int dz = t % LD; const int c = t / LD;
        const int id = id0 + dz, ih = ih0 + hy, ci = c0 + c;
        const bool row_ok = (id >= 0) && (id < d.ID) && (ih >= 0) && (ih < d.IH);
        const float* src = x + (((size_t)n * d.CI + ci) * d.ID + (row_ok ? id : 0)) * (size_t)d.IH * d.IW
                             + (size_t)(row_ok ? ih : 0) * d.IW;
        float sc = 1.f, sh = 0.f;
        if (in_scale != nullptr) { sc = in_scale[g * d.CI + ci]; sh = in_shift[g * d.CI + ci]; }
        float* dst = lds + (size_t)r * LWp;
        for (int wx = lane; wx < LW; wx += VG_WAVE) {
            const int iw = iw0 + wx;
            float v = 0.f;
            if (row_ok && iw >= 0 && iw < d.IW) {
                v = src[iw];
                if (d.relu_in) v = fmaxf(v, 0.f);
                v = fmaf(v, sc, sh);
            }
            dst[wx] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// corr3d
// ------------------------------------------------------------------------------------------
template <int CO, int KD, int KH, int KW, int S, int TDt, int THt, int TW>
__global__ void __launch_bounds__(256)
corr3d_k(const float* __restrict__ x, const float* __restrict__ wpk, const float* __restrict__ bias,
         const float* __restrict__ in_scale, const float* __restrict__ in_shift,
         const float* __restrict__ mask_src, float* __restrict__ y, CorrParams p) {
    VG_DYN_SMEM(float, lds);
    constexpr int KVOL = KD * KH * KW;
    constexpr int RW = (TW - 1) * S + KW;
    constexpr int RD = (TDt - 1) * S + KD;
    constexpr int RH = (THt - 1) * S + KH;
    const vg_conv_desc& d = p.d;
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    int tile = blockIdx.x;
    const int twi = tile % p.tilesW; tile /= p.tilesW;
    const int thi = tile % p.tilesH; const int tdi = tile / p.tilesH;
    const int wg = tid % p.TWG; const int thl = (tid / p.TWG) % p.TH; const int tdl = tid / (p.TWG * p.TH);
    const bool active = tdl < p.TD;
    const int od0 = tdi * p.TD * TDt, oh0 = thi * p.TH * THt, ow0 = twi * p.TWG * TW;
    const int id0 = od0 * S - d.pad_d, ih0 = oh0 * S - d.pad_h, iw0 = ow0 * S - d.pad_w;

    float acc[TDt][THt][TW][CO];
#pragma unroll
    for (int a = 0; a < TDt; ++a)
#pragma unroll
        for (int b = 0; b < THt; ++b)
#pragma unroll
            for (int j = 0; j < TW; ++j)
#pragma unroll
                for (int co = 0; co < CO; ++co) acc[a][b][j][co] = 0.f;

    for (int c0 = 0; c0 < d.CI; c0 += p.CCH) {
        const int cc = min(p.CCH, d.CI - c0);
        __syncthreads();
        stage_tile(lds, x, in_scale, in_shift, d, n, c0, cc, id0, ih0, iw0, p.LD, p.LH, p.LW, p.LWp);
        __syncthreads();
        if (active) {
            for (int c = 0; c < cc; ++c) {
                const float* __restrict__ wc = wpk + (size_t)(c0 + c) * KVOL * CO;
                const float* tl = lds + ((size_t)(c * p.LD + tdl * TDt * S) * p.LH + thl * THt * S) * p.LWp + wg * TW * S;
#pragma unroll
                for (int dz = 0; dz < RD; ++dz) {
#pragma unroll
                    for (int hy = 0; hy < RH; ++hy) {
                        float seg[RW];
                        const float* row = tl + ((size_t)dz * p.LH + hy) * p.LWp;
#pragma unroll
                        for (int i = 0; i < RW; ++i) seg[i] = row[i];
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
                                    for (int co = 0; co < CO; ++co) {
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
    const size_t plane = (size_t)d.OH * d.OW;
#pragma unroll
    for (int a = 0; a < TDt; ++a) {
        const int od = od0 + tdl * TDt + a;
        if (od >= d.OD) continue;
#pragma unroll
        for (int b = 0; b < THt; ++b) {
            const int oh = oh0 + thl * THt + b;
            if (oh >= d.OH) continue;
#pragma unroll
            for (int co = 0; co < CO; ++co) {
                const float bv = bias ? bias[co] : 0.f;
                const size_t base = (((size_t)n * CO + co) * d.OD + od) * plane + (size_t)oh * d.OW;
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    const int ow = ow0 + wg * TW + j;
                    if (ow < d.OW) {
                        float v = acc[a][b][j][co] + bv;
                        if (mask_src) v = (mask_src[base + ow] > 0.f) ? v : 0.f;
                        y[base + ow] = v;
                    }
                }
            }
        }
    }
}

template <int CO, int KD, int KH, int KW, int S, int TDt, int THt, int TW>
int launch_corr(const vg_conv_desc* d, const float* x, const float* wpk, const float* bias, const float* in_scale,
                const float* in_shift, const float* mask_src, float* y, hipStream_t s) {
    CorrParams p; p.d = *d;
    const int gw = vg_cdiv(d->OW, TW), gh = vg_cdiv(d->OH, THt), gd = vg_cdiv(d->OD, TDt);
    p.TWG = gw < 16 ? gw : 16;
    int rem = 256 / p.TWG;
    p.TH = gh < rem ? gh : rem;
    rem = 256 / (p.TWG * p.TH);
    p.TD = gd < rem ? gd : rem;
    if (p.TD < 1) p.TD = 1;
    p.tilesW = vg_cdiv(gw, p.TWG); p.tilesH = vg_cdiv(gh, p.TH); p.tilesD = vg_cdiv(gd, p.TD);
    p.LW = (p.TWG * TW - 1) * S + KW; p.LH = (p.TH * THt - 1) * S + KH; p.LD = (p.TD * TDt - 1) * S + KD;
    p.LWp = p.LW | 1;                               // odd row pitch: rows interleave over the LDS banks
    if ((p.LWp & 3) == 3) p.LWp += 2;               // pitch = 1 (mod 4)
    const size_t per_ch = (size_t)p.LD * p.LH * p.LWp * sizeof(float);
    int cch = (int)((size_t)40960 / per_ch);
    if (cch < 1) cch = 1;
    if (cch > d->CI) cch = d->CI;
    p.CCH = cch;
    const size_t shmem = per_ch * cch;
    if (shmem > 64 * 1024) { vg_set_error("corr3d: LDS tile of %zu bytes too large", shmem); return VG_ERR_UNSUPPORTED; }
    const int threads = vg_cdiv(p.TWG * p.TH * p.TD, VG_WAVE) * VG_WAVE;
    dim3 grid(p.tilesW * p.tilesH * p.tilesD, d->N);
    vg_launch(corr3d_k<CO, KD, KH, KW, S, TDt, THt, TW>, grid, dim3(threads), shmem, s,
              x, wpk, bias, in_scale, in_shift, mask_src, y, p);
    return vg_check_launch("corr3d");
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
    int LD, LH, LW, LWp, CCH;
};

template <int CO, int KD, int KH, int KW>
__global__ void __launch_bounds__(256)
tconv3d_s2_k(const float* __restrict__ x, const float* __restrict__ wpk, const float* __restrict__ bias,
             const float* __restrict__ in_scale, const float* __restrict__ in_shift,
             const float* __restrict__ mask_src, float* __restrict__ y, TconvParams p) {
    VG_DYN_SMEM(float, lds);
    constexpr int KVOL = KD * KH * KW;
    constexpr int MD = (KD + 1) / 2, MH = (KH + 1) / 2, MW = (KW + 1) / 2;
    const vg_conv_desc& d = p.d;
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    int tile = blockIdx.x;
    const int twi = tile % p.tilesW; tile /= p.tilesW;
    const int thi = tile % p.tilesH; const int tdi = tile / p.tilesH;
    const int jwl = tid % p.TJW; const int jhl = (tid / p.TJW) % p.TJH; const int jdl = tid / (p.TJW * p.TJH);
    const bool active = jdl < p.TJD;
    const int jd0 = tdi * p.TJD, jh0 = thi * p.TJH, jw0 = twi * p.TJW;
    // LDS tile origin in input coordinates: i = j0 - (M-1)
    const int id0 = jd0 - (MD - 1), ih0 = jh0 - (MH - 1), iw0 = jw0 - (MW - 1);

    float acc[2][2][2][CO];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int co = 0; co < CO; ++co) acc[a][b][c][co] = 0.f;

    for (int c0 = 0; c0 < d.CI; c0 += p.CCH) {
        const int cc = min(p.CCH, d.CI - c0);
        __syncthreads();
        stage_tile(lds, x, in_scale, in_shift, d, n, c0, cc, id0, ih0, iw0, p.LD, p.LH, p.LW, p.LWp);
        __syncthreads();
        if (active) {
            for (int c = 0; c < cc; ++c) {
                const float* __restrict__ wc = wpk + (size_t)(c0 + c) * KVOL * CO;
                // local coordinate of input i = j - m is (jl - m + M - 1)
                const float* tl = lds + ((size_t)(c * p.LD + jdl) * p.LH + jhl) * p.LWp + jwl;
#pragma unroll
                for (int md = 0; md < MD; ++md)
#pragma unroll
                    for (int mh = 0; mh < MH; ++mh)
#pragma unroll
                        for (int mw = 0; mw < MW; ++mw) {
                            const float xv = tl[((size_t)(MD - 1 - md) * p.LH + (MH - 1 - mh)) * p.LWp + (MW - 1 - mw)];
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
                                        for (int co = 0; co < CO; ++co)
                                            acc[rd][rh][rw][co] = fmaf(xv, wc[t * CO + co], acc[rd][rh][rw][co]);
                                    }
                                }
                            }
                        }
            }
        }
    }
    if (!active) return;
    const int jd = jd0 + jdl, jh = jh0 + jhl, jw = jw0 + jwl;
    const size_t plane = (size_t)d.OH * d.OW;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int od = 2 * jd + rd - d.pad_d;
        if (od < 0 || od >= d.OD) continue;
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
            const int oh = 2 * jh + rh - d.pad_h;
            if (oh < 0 || oh >= d.OH) continue;
#pragma unroll
            for (int co = 0; co < CO; ++co) {
                const float bv = bias ? bias[co] : 0.f;
                const size_t base = (((size_t)n * CO + co) * d.OD + od) * plane + (size_t)oh * d.OW;
#pragma unroll
                for (int rw = 0; rw < 2; ++rw) {
                    const int ow = 2 * jw + rw - d.pad_w;
                    if (ow >= 0 && ow < d.OW) {
                        float v = acc[rd][rh][rw][co] + bv;
                        if (mask_src) v = (mask_src[base + ow] > 0.f) ? v : 0.f;
                        y[base + ow] = v;
                    }
                }
            }
        }
    }
}

template <int CO, int KD, int KH, int KW>
int launch_tconv(const vg_conv_desc* d, const float* x, const float* wpk, const float* bias, const float* in_scale,
                 const float* in_shift, const float* mask_src, float* y, hipStream_t s) {
    constexpr int MD = (KD + 1) / 2, MH = (KH + 1) / 2, MW = (KW + 1) / 2;
    TconvParams p; p.d = *d;
    p.JD = (d->OD + d->pad_d + 1) / 2; p.JH = (d->OH + d->pad_h + 1) / 2; p.JW = (d->OW + d->pad_w + 1) / 2;
    p.TJW = p.JW < 32 ? p.JW : 32;
    int rem = 256 / p.TJW;
    p.TJH = p.JH < rem ? p.JH : rem;
    rem = 256 / (p.TJW * p.TJH);
    p.TJD = p.JD < rem ? p.JD : rem;
    if (p.TJD < 1) p.TJD = 1;
    p.tilesW = vg_cdiv(p.JW, p.TJW); p.tilesH = vg_cdiv(p.JH, p.TJH); p.tilesD = vg_cdiv(p.JD, p.TJD);
    p.LW = p.TJW + MW - 1; p.LH = p.TJH + MH - 1; p.LD = p.TJD + MD - 1;
    p.LWp = p.LW | 1;
    const size_t per_ch = (size_t)p.LD * p.LH * p.LWp * sizeof(float);
    int cch = (int)((size_t)40960 / per_ch);
    if (cch < 1) cch = 1;
    if (cch > d->CI) cch = d->CI;
    p.CCH = cch;
    const size_t shmem = per_ch * cch;
    if (shmem > 64 * 1024) { vg_set_error("tconv3d_s2: LDS tile of %zu bytes too large", shmem); return VG_ERR_UNSUPPORTED; }
    const int threads = vg_cdiv(p.TJW * p.TJH * p.TJD, VG_WAVE) * VG_WAVE;
    dim3 grid(p.tilesW * p.tilesH * p.tilesD, d->N);
    vg_launch(tconv3d_s2_k<CO, KD, KH, KW>, grid, dim3(threads), shmem, s,
              x, wpk, bias, in_scale, in_shift, mask_src, y, p);
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
    const int key = d->CO * 100000 + d->KD * 10000 + d->KH * 1000 + d->KW * 100 + d->stride;
#define CORR_CASE(CO, KD, KH, KW, S, TDt, THt, TW) \
    case CO * 100000 + KD * 10000 + KH * 1000 + KW * 100 + S: \
        return launch_corr<CO, KD, KH, KW, S, TDt, THt, TW>(d, x, wpk, bias, in_scale, in_shift, mask_src, y, s);
    switch (key) {
        CORR_CASE(1, 3, 3, 3, 1, 2, 2, 4)
        CORR_CASE(8, 3, 3, 3, 1, 1, 1, 4)
        CORR_CASE(16, 3, 3, 3, 1, 1, 1, 4)
        CORR_CASE(8, 3, 3, 3, 2, 1, 1, 4)
        CORR_CASE(16, 3, 3, 3, 2, 1, 1, 2)
        CORR_CASE(8, 5, 3, 3, 2, 1, 1, 4)
        CORR_CASE(8, 4, 4, 4, 2, 1, 1, 4)
        default: break;
    }
#undef CORR_CASE
    vg_set_error("vg_corr3d: no kernel instance for CO=%d k=%dx%dx%d stride=%d", d->CO, d->KD, d->KH, d->KW, d->stride);
    return VG_ERR_UNSUPPORTED;
}

extern "C" int vg_tconv3d_s2(const vg_conv_desc* d, const float* x, const float* wpk, const float* bias,
                             const float* in_scale, const float* in_shift, const float* mask_src, float* y, void* stream) {
    int rc = check_desc(d, x, wpk, y, "vg_tconv3d_s2");
    if (rc) return rc;
    if (d->stride != 2) { vg_set_error("vg_tconv3d_s2: stride must be 2"); return VG_ERR_ARG; }
    if ((in_scale == nullptr) != (in_shift == nullptr) || (in_scale && d->per_group <= 0)) {
        vg_set_error("vg_tconv3d_s2: in_scale/in_shift/per_group inconsistent"); return VG_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const int key = d->CO * 1000 + d->KD * 100 + d->KH * 10 + d->KW;
#define TCONV_CASE(CO, KD, KH, KW) \
    case CO * 1000 + KD * 100 + KH * 10 + KW: \
        return launch_tconv<CO, KD, KH, KW>(d, x, wpk, bias, in_scale, in_shift, mask_src, y, s);
    switch (key) {
        TCONV_CASE(8, 3, 3, 3)
        TCONV_CASE(16, 3, 3, 3)
        TCONV_CASE(8, 5, 3, 3)
        TCONV_CASE(8, 4, 4, 4)
        default: break;
    }
#undef TCONV_CASE
    vg_set_error("vg_tconv3d_s2: no kernel instance for CO=%d k=%dx%dx%d", d->CO, d->KD, d->KH, d->KW);
    return VG_ERR_UNSUPPORTED;
}
