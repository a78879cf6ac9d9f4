// vg_wgrad.hip -- convolution weight gradients (gfx950).
//
//   dw[cb][ca][k] = sum_{n,p} PB(b)[n][cb][p] * PA(a)[n][ca][p*S + k - pad]
//
// The output is tiny (<= 16*16*45 values) and the reduction runs over every position of every
// sample, so blocks are persistent: each walks a strided list of (sample, position-tile) work
// items keeping its share of dw in registers, writes ONE partial slab at the end, and a second
// kernel sums the slabs in a fixed order (deterministic; no float atomics).
//
//   "owner" variant : a thread owns dw[cb0..cb0+CBT)[ca][kd][(kh)][*]; tiles of a and b are staged in
//                     LDS (b as [p][cb] so one ds_read_b128 feeds CBT accumulator columns).
//   "wide" variant  : CB*CA*KVOL <= 216 (conv1 / convt5, one side has a single channel): every
//                     thread owns ALL of dw for its own positions; wave-shuffle + LDS reduction.
#include "vg_common.h"
#include "../../include/vaegam.h"

namespace {

struct WgradParams {
    vg_wgrad_desc d;
    int TPD, TPH;               // position tile (TPW is a template parameter)
    int tilesW, tilesH, tilesD;
    int LD, LH, LW, LWp;        // a-tile geometry (per channel)
    int items;                  // N * tiles
    int psplit;                 // owner sets per block
    int b_off;                  // float offset of the b tile inside LDS
};

__device__ __forceinline__ float apply_pro(float v, int relu, float sc, float sh) {
    if (relu) v = fmaxf(v, 0.f);
    return fmaf(v, sc, sh);
}

template <int CB, int CBT, int KD, int KH, int KW, int S, int TPW, bool OWN_KH>
__global__ void __launch_bounds__(256)
wgrad_own_k(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ in_scale,
            const float* __restrict__ in_shift, float* __restrict__ ws, WgradParams p) {
    VG_DYN_SMEM(float, lds);
    constexpr int KVOL = KD * KH * KW;
    constexpr int NKH = OWN_KH ? 1 : KH;
    const vg_wgrad_desc& d = p.d;
    const int CA = d.CA;
    const int owners = (CB / CBT) * CA * KD * (OWN_KH ? KH : 1);
    const int tid = threadIdx.x;
    const int ps = tid / owners;                 // which position sub-set this thread accumulates
    const int ow_id = tid % owners;
    const bool active = ps < p.psplit;
    // owner id -> (cbq, ca, kd[, kh])
    int o = ow_id;
    int kh_own = 0;
    if (OWN_KH) { kh_own = o % KH; o /= KH; }
    const int kd = o % KD; o /= KD;
    const int ca = o % CA; const int cbq = o / CA;

    float acc[NKH][KW][CBT];
#pragma unroll
    for (int i = 0; i < NKH; ++i)
#pragma unroll
        for (int j = 0; j < KW; ++j)
#pragma unroll
            for (int c = 0; c < CBT; ++c) acc[i][j][c] = 0.f;

    float* atile = lds;
    float* btile = lds + p.b_off;
    const int lane = tid % VG_WAVE, wave = vg_wave_id(), nwaves = blockDim.x / VG_WAVE;
    const int tiles = p.tilesW * p.tilesH * p.tilesD;

    for (int item = blockIdx.x; item < p.items; item += gridDim.x) {
        const int n = item / tiles; int tile = item % tiles;
        const int twi = tile % p.tilesW; tile /= p.tilesW;
        const int thi = tile % p.tilesH; const int tdi = tile / p.tilesH;
        const int pd0 = tdi * p.TPD, ph0 = thi * p.TPH, pw0 = twi * TPW;
        const int g = (in_scale != nullptr) ? n / d.per_group : 0;
        __syncthreads();
        // ---- stage a tile: [CA][LD][LH][LWp], zero outside, prologue if it belongs to a.
        // A wave takes groups of U consecutive rows (U independent global loads in flight); the row's
        // (c, dz, hy) is carried incrementally: no scalar div/mod per row.
        {
            constexpr int U = 4;
            const int ad0 = pd0 * S - d.pad_d, ah0 = ph0 * S - d.pad_h, aw0 = pw0 * S - d.pad_w;
            const int rows = CA * p.LD * p.LH;
            const size_t aplane = (size_t)d.AH * d.AW;
            const int rl = d.pro_on_a ? d.relu_in : 0;
            int c = 0, dz = 0, hy = wave * U;
            while (hy >= p.LH) { hy -= p.LH; if (++dz == p.LD) { dz = 0; ++c; } }
            for (int r0 = wave * U; r0 < rows; r0 += nwaves * U) {
                const float* src[U]; float sc[U], sh[U]; bool ok[U];
                int cu = c, dzu = dz, hyu = hy;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int cc_ = cu < CA ? cu : 0;
                    const int id = ad0 + dzu, ih = ah0 + hyu;
                    ok[u] = r0 + u < rows && id >= 0 && id < d.AD && ih >= 0 && ih < d.AH;
                    src[u] = a + (((size_t)n * CA + cc_) * d.AD + (ok[u] ? id : 0)) * aplane + (size_t)(ok[u] ? ih : 0) * d.AW;
                    sc[u] = 1.f; sh[u] = 0.f;
                    if (d.pro_on_a && in_scale) { sc[u] = in_scale[g * CA + cc_]; sh[u] = in_shift[g * CA + cc_]; }
                    if (++hyu == p.LH) { hyu = 0; if (++dzu == p.LD) { dzu = 0; ++cu; } }
                }
                for (int wx = lane; wx < p.LW; wx += VG_WAVE) {
                    const int iw = aw0 + wx;
                    const bool cok = iw >= 0 && iw < d.AW;
                    const int iwc = min(max(iw, 0), d.AW - 1);       // valid address: unconditional loads stay in flight
                    float v[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) v[u] = src[u][iwc];
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (r0 + u < rows) atile[(size_t)(r0 + u) * p.LWp + wx] = (ok[u] && cok) ? apply_pro(v[u], rl, sc[u], sh[u]) : 0.f;
                }
                hy += nwaves * U;
                while (hy >= p.LH) { hy -= p.LH; if (++dz == p.LD) { dz = 0; ++c; } }
            }
        }
        // ---- stage b tile: [TPD][TPH][TPW][CB] (cb fastest), zero outside; rows ordered (c, dz, hy)
        {
            constexpr int U = 4;
            const int rows = CB * p.TPD * p.TPH;
            const size_t bplane = (size_t)d.PH * d.PW;
            const int rl = d.pro_on_a ? 0 : d.relu_in;
            int c = 0, dz = 0, hy = wave * U;
            while (hy >= p.TPH) { hy -= p.TPH; if (++dz == p.TPD) { dz = 0; ++c; } }
            for (int r0 = wave * U; r0 < rows; r0 += nwaves * U) {
                const float* src[U]; float sc[U], sh[U]; bool ok[U]; int dsto[U];
                int cu = c, dzu = dz, hyu = hy;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int cc_ = cu < CB ? cu : 0;
                    const int pd = pd0 + dzu, ph = ph0 + hyu;
                    ok[u] = r0 + u < rows && pd < d.PD && ph < d.PH;
                    src[u] = b + (((size_t)n * CB + cc_) * d.PD + (ok[u] ? pd : 0)) * bplane + (size_t)(ok[u] ? ph : 0) * d.PW;
                    sc[u] = 1.f; sh[u] = 0.f;
                    if (!d.pro_on_a && in_scale) { sc[u] = in_scale[g * CB + cc_]; sh[u] = in_shift[g * CB + cc_]; }
                    dsto[u] = (dzu * p.TPH + hyu) * TPW * CB + cc_;
                    if (++hyu == p.TPH) { hyu = 0; if (++dzu == p.TPD) { dzu = 0; ++cu; } }
                }
                for (int wx = lane; wx < TPW; wx += VG_WAVE) {
                    const int pw = pw0 + wx;
                    const bool cok = pw < d.PW;
                    const int pwc = min(pw, d.PW - 1);
                    float v[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) v[u] = src[u][pwc];
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (r0 + u < rows) btile[(size_t)dsto[u] + (size_t)wx * CB] = (ok[u] && cok) ? apply_pro(v[u], rl, sc[u], sh[u]) : 0.f;
                }
                hy += nwaves * U;
                while (hy >= p.TPH) { hy -= p.TPH; if (++dz == p.TPD) { dz = 0; ++c; } }
            }
        }
        __syncthreads();
        if (active) {
            for (int pd = ps; pd < p.TPD; pd += p.psplit) {
                for (int ph = 0; ph < p.TPH; ++ph) {
                    const float* brow = btile + (size_t)((pd * p.TPH + ph) * TPW) * CB + cbq * CBT;
                    const float* arow0 = atile + ((size_t)(ca * p.LD + pd * S + kd) * p.LH + ph * S + kh_own) * p.LWp;
#pragma unroll
                    for (int pw = 0; pw < TPW; ++pw) {
                        float bv[CBT];
                        if (CBT == 4) {                       // b tile is [p][cb], 16-byte aligned groups of 4 channels
                            const float4 t4 = *reinterpret_cast<const float4*>(brow + pw * CB);
                            bv[0] = t4.x; bv[1] = t4.y; bv[2] = t4.z; bv[3] = t4.w;
                        } else {
#pragma unroll
                            for (int c = 0; c < CBT; ++c) bv[c] = brow[pw * CB + c];
                        }
#pragma unroll
                        for (int i = 0; i < NKH; ++i) {
#pragma unroll
                            for (int j = 0; j < KW; ++j) {
                                const float av = arow0[(size_t)i * p.LWp + pw * S + j];
#pragma unroll
                                for (int c = 0; c < CBT; ++c) acc[i][j][c] = fmaf(av, bv[c], acc[i][j][c]);
                            }
                        }
                    }
                }
            }
        }
    }
    // ---- one partial slab per (block, ps): ws[(block*psplit + ps)][cb][ca][k]
    if (active) {
        float* out = ws + ((size_t)blockIdx.x * p.psplit + ps) * (size_t)CB * CA * KVOL;
#pragma unroll
        for (int i = 0; i < NKH; ++i) {
            const int kh = OWN_KH ? kh_own : i;
#pragma unroll
            for (int j = 0; j < KW; ++j)
#pragma unroll
                for (int c = 0; c < CBT; ++c)
                    out[((size_t)(cbq * CBT + c) * CA + ca) * KVOL + (kd * KH + kh) * KW + j] = acc[i][j][c];
        }
    }
}

// sum `nslab` slabs of `len` floats in slab order (fixed order => run-to-run reproducible)
__global__ void __launch_bounds__(256) slab_sum_k(const float* __restrict__ ws, int nslab, int len, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < nslab; k += 4) {
        s0 += ws[(size_t)k * len + i]; s1 += ws[(size_t)(k + 1) * len + i];
        s2 += ws[(size_t)(k + 2) * len + i]; s3 += ws[(size_t)(k + 3) * len + i];
    }
    for (; k < nslab; ++k) s0 += ws[(size_t)k * len + i];
    out[i] = (s0 + s1) + (s2 + s3);
}

constexpr int WG_MAX_BLOCKS = 512;

template <int CB, int CBT, int KD, int KH, int KW, int S, int TPW, bool OWN_KH>
int plan_own(const vg_wgrad_desc* d, WgradParams& p, size_t& shmem, int& threads, int& grid) {
    p.d = *d;
    const int owners = (CB / CBT) * d->CA * KD * (OWN_KH ? KH : 1);
    if (owners > 256) { vg_set_error("wgrad: %d owners exceed a block", owners); return VG_ERR_UNSUPPORTED; }
    p.psplit = 256 / owners;
    p.LW = (TPW - 1) * S + KW; p.LWp = p.LW | 1;
    // grow the position tile (h first, then d) while a + b tiles fit 40 KiB
    int best_h = 1, best_d = 1;
    for (int td = 1; td <= 8; ++td)
        for (int th = 1; th <= 8; ++th) {
            if (th > d->PH || td > d->PD) continue;
            const size_t fl = (size_t)d->CA * ((td - 1) * S + KD) * ((th - 1) * S + KH) * p.LWp + (size_t)td * th * TPW * CB;
            if (fl * 4 <= 40960 && td * th > best_d * best_h) { best_d = td; best_h = th; }
        }
    p.TPD = best_d; p.TPH = best_h;
    if (p.psplit > p.TPD) p.psplit = p.TPD;          // sub-sets split the tile along d
    p.LD = (p.TPD - 1) * S + KD; p.LH = (p.TPH - 1) * S + KH;
    p.tilesW = vg_cdiv(d->PW, TPW); p.tilesH = vg_cdiv(d->PH, p.TPH); p.tilesD = vg_cdiv(d->PD, p.TPD);
    p.items = d->N * p.tilesW * p.tilesH * p.tilesD;
    const size_t afl = (size_t)d->CA * p.LD * p.LH * p.LWp;
    p.b_off = (int)((afl + 3) & ~(size_t)3);
    shmem = ((size_t)p.b_off + (size_t)p.TPD * p.TPH * TPW * CB) * sizeof(float);
    if (shmem > 64 * 1024) { vg_set_error("wgrad: LDS tile of %zu bytes too large", shmem); return VG_ERR_UNSUPPORTED; }
    threads = vg_cdiv(owners * p.psplit, VG_WAVE) * VG_WAVE;
    grid = p.items < WG_MAX_BLOCKS ? p.items : WG_MAX_BLOCKS;
    return VG_OK;
}

template <int CB, int CBT, int KD, int KH, int KW, int S, int TPW, bool OWN_KH>
int launch_own(const vg_wgrad_desc* d, const float* a, const float* b, const float* in_scale, const float* in_shift,
               float* ws, float* dw, hipStream_t s, int64_t* ws_bytes_only) {
    WgradParams p; size_t shmem; int threads, grid;
    int rc = plan_own<CB, CBT, KD, KH, KW, S, TPW, OWN_KH>(d, p, shmem, threads, grid);
    if (rc) return rc;
    const int len = CB * d->CA * KD * KH * KW;
    if (ws_bytes_only) { *ws_bytes_only = (int64_t)grid * p.psplit * len * sizeof(float); return VG_OK; }
    vg_launch(wgrad_own_k<CB, CBT, KD, KH, KW, S, TPW, OWN_KH>, dim3(grid), dim3(threads), shmem, s,
              a, b, in_scale, in_shift, ws, p);
    rc = vg_check_launch("wgrad_own");
    if (rc) return rc;
    vg_launch(slab_sum_k, dim3(vg_cdiv(len, 256)), dim3(256), 0, s, (const float*)ws, grid * p.psplit, len, dw);
    return vg_check_launch("wgrad slab_sum");
}

// ------------------------------------------------------------------------------------------
// wide variant: CB = 8, CA = 1, 3x3x3, stride 1, pad 0  (conv1 and convt5)
// ------------------------------------------------------------------------------------------
struct WideParams { vg_wgrad_desc d; int wgroups; long long items; };

template <int CB, int KD, int KH, int KW, int TW>
__global__ void __launch_bounds__(256)
wgrad_wide_k(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ in_scale,
             const float* __restrict__ in_shift, float* __restrict__ ws, WideParams p) {
    constexpr int KVOL = KD * KH * KW;
    constexpr int RW = TW - 1 + KW;
    __shared__ float red[4][CB * KVOL];
    const vg_wgrad_desc& d = p.d;
    float acc[CB][KVOL];
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int k = 0; k < KVOL; ++k) acc[c][k] = 0.f;

    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < p.items; it += stride) {
        long long r = it;
        const int wgp = (int)(r % p.wgroups); r /= p.wgroups;
        const int ph = (int)(r % d.PH); r /= d.PH;
        const int pd = (int)(r % d.PD); const int n = (int)(r / d.PD);
        const int pw0 = wgp * TW;
        const int g = (in_scale != nullptr) ? n / d.per_group : 0;
        float bv[CB][TW];
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const float* src = b + ((((size_t)n * CB + c) * d.PD + pd) * d.PH + ph) * (size_t)d.PW;
            float sc = 1.f, sh = 0.f; int rl = 0;
            if (!d.pro_on_a) { rl = d.relu_in; if (in_scale) { sc = in_scale[g * CB + c]; sh = in_shift[g * CB + c]; } }
#pragma unroll
            for (int j = 0; j < TW; ++j) {                   // clamped address + select: loads stay unconditional
                const float t = src[min(pw0 + j, d.PW - 1)];
                bv[c][j] = (pw0 + j < d.PW) ? apply_pro(t, rl, sc, sh) : 0.f;
            }
        }
        float sca = 1.f, sha = 0.f; int rla = 0;
        if (d.pro_on_a) { rla = d.relu_in; if (in_scale) { sca = in_scale[g]; sha = in_shift[g]; } }
#pragma unroll
        for (int kd = 0; kd < KD; ++kd)
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) {
                const float* src = a + (((size_t)n * d.AD + (pd + kd)) * d.AH + (ph + kh)) * (size_t)d.AW;
                float seg[RW];
#pragma unroll
                for (int i = 0; i < RW; ++i) {
                    const float t = src[min(pw0 + i, d.AW - 1)];
                    seg[i] = (pw0 + i < d.AW) ? apply_pro(t, rla, sca, sha) : 0.f;
                }
#pragma unroll
                for (int kw = 0; kw < KW; ++kw)
#pragma unroll
                    for (int c = 0; c < CB; ++c)
#pragma unroll
                        for (int j = 0; j < TW; ++j)
                            acc[c][(kd * KH + kh) * KW + kw] = fmaf(seg[j + kw], bv[c][j], acc[c][(kd * KH + kh) * KW + kw]);
            }
    }
    // wave reduction (64 lanes), then the 4 waves through LDS
    const int lane = threadIdx.x % VG_WAVE, wave = threadIdx.x / VG_WAVE;
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int k = 0; k < KVOL; ++k) {
            float v = acc[c][k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0) red[wave][c * KVOL + k] = v;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < CB * KVOL; i += blockDim.x)
        ws[(size_t)blockIdx.x * CB * KVOL + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

int launch_wide(const vg_wgrad_desc* d, const float* a, const float* b, const float* in_scale, const float* in_shift,
                float* ws, float* dw, hipStream_t s, int64_t* ws_bytes_only) {
    constexpr int TW = 4;
    WideParams p; p.d = *d; p.wgroups = vg_cdiv(d->PW, TW);
    p.items = (long long)d->N * d->PD * d->PH * p.wgroups;
    long long want = (p.items + 255) / 256;
    const int grid = (int)(want < WG_MAX_BLOCKS ? want : WG_MAX_BLOCKS);
    const int len = 8 * 27;
    if (ws_bytes_only) { *ws_bytes_only = (int64_t)grid * len * sizeof(float); return VG_OK; }
    vg_launch(wgrad_wide_k<8, 3, 3, 3, TW>, dim3(grid), dim3(256), 0, s, a, b, in_scale, in_shift, ws, p);
    int rc = vg_check_launch("wgrad_wide");
    if (rc) return rc;
    vg_launch(slab_sum_k, dim3(vg_cdiv(len, 256)), dim3(256), 0, s, (const float*)ws, grid, len, dw);
    return vg_check_launch("wgrad slab_sum");
}

int dispatch(const vg_wgrad_desc* d, const float* a, const float* b, const float* in_scale, const float* in_shift,
             float* ws, float* dw, hipStream_t s, int64_t* ws_only) {
    if (!d) { vg_set_error("vg_wgrad3d: null descriptor"); return VG_ERR_ARG; }
    if (d->N <= 0 || d->CA <= 0 || d->CB <= 0 || d->PD <= 0 || d->PH <= 0 || d->PW <= 0 || d->AD <= 0 || d->AH <= 0 ||
        d->AW <= 0 || (d->stride != 1 && d->stride != 2)) {
        vg_set_error("vg_wgrad3d: bad shape"); return VG_ERR_ARG;
    }
    if ((in_scale == nullptr) != (in_shift == nullptr) || (in_scale && d->per_group <= 0)) {
        vg_set_error("vg_wgrad3d: in_scale/in_shift/per_group inconsistent"); return VG_ERR_ARG;
    }
    const bool k333 = d->KD == 3 && d->KH == 3 && d->KW == 3;
    const bool small_w = d->PW <= 8;
    if (d->CB == 8 && d->CA == 1 && k333 && d->stride == 1 && d->pad_d == 0 && d->pad_h == 0 && d->pad_w == 0 &&
        d->AD >= d->PD + 2 && d->AH >= d->PH + 2)
        return launch_wide(d, a, b, in_scale, in_shift, ws, dw, s, ws_only);
#define OWN(CB, CBT, KD, KH, KW, S, TPW, OKH) \
    return launch_own<CB, CBT, KD, KH, KW, S, TPW, OKH>(d, a, b, in_scale, in_shift, ws, dw, s, ws_only)
    if (k333 && d->CB == 16 && d->CA == 16 && d->stride == 1) { if (small_w) OWN(16, 4, 3, 3, 3, 1, 8, false); OWN(16, 4, 3, 3, 3, 1, 16, false); }
    if (k333 && d->CB == 16 && d->CA == 16 && d->stride == 2) { if (small_w) OWN(16, 4, 3, 3, 3, 2, 8, false); OWN(16, 4, 3, 3, 3, 2, 16, false); }
    if (k333 && d->CB == 16 && d->CA == 8 && d->stride == 1) { if (small_w) OWN(16, 4, 3, 3, 3, 1, 8, false); OWN(16, 4, 3, 3, 3, 1, 16, false); }
    if (k333 && d->CB == 8 && d->CA == 8 && d->stride == 2) { if (small_w) OWN(8, 4, 3, 3, 3, 2, 8, true); OWN(8, 4, 3, 3, 3, 2, 16, true); }
    if (k333 && d->CB == 8 && d->CA == 8 && d->stride == 1) { if (small_w) OWN(8, 4, 3, 3, 3, 1, 8, true); OWN(8, 4, 3, 3, 3, 1, 16, true); }
    if (d->KD == 5 && d->KH == 3 && d->KW == 3 && d->CB == 8 && d->CA == 8 && d->stride == 2) OWN(8, 4, 5, 3, 3, 2, 16, true);
    if (d->KD == 4 && d->KH == 4 && d->KW == 4 && d->CB == 8 && d->CA == 8 && d->stride == 2) OWN(8, 4, 4, 4, 4, 2, 16, true);
#undef OWN
    vg_set_error("vg_wgrad3d: no kernel instance for CB=%d CA=%d k=%dx%dx%d stride=%d", d->CB, d->CA, d->KD, d->KH, d->KW, d->stride);
    return VG_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int64_t vg_wgrad3d_ws_bytes(const vg_wgrad_desc* d) {
    int64_t bytes = 0;
    int rc = dispatch(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &bytes);
    return rc ? -1 : bytes;
}

extern "C" int vg_wgrad3d(const vg_wgrad_desc* d, const float* a, const float* b, const float* in_scale,
                          const float* in_shift, float* ws, float* dw, void* stream) {
    if (!a || !b || !ws || !dw) { vg_set_error("vg_wgrad3d: null argument"); return VG_ERR_ARG; }
    return dispatch(d, a, b, in_scale, in_shift, ws, dw, (hipStream_t)stream, nullptr);
}
