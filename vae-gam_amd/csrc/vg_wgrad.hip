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
#include <stdlib.h>
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

__device__ __forceinline__ int clampi_(int v, int lo, int hi) { return min(max(v, lo), hi); }

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

// sum `nslab` slabs of `len` floats in a fixed order (run-to-run reproducible): a block owns 64 outputs, its 16
// thread rows take slabs r, r+16, ... (4 running sums each), LDS tree over the rows; optionally += into out
constexpr int SLAB_ROWS = 16;
__global__ void __launch_bounds__(64 * SLAB_ROWS)
slab_sum_k(const float* __restrict__ ws, int nslab, int len, int accumulate, float* __restrict__ out) {
    __shared__ float red[SLAB_ROWS][64];
    const int j = threadIdx.x % 64, r = threadIdx.x / 64;
    const int i = blockIdx.x * 64 + j;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < len) {
        int k = r;
        for (; k + 3 * SLAB_ROWS < nslab; k += 4 * SLAB_ROWS) {
            s0 += ws[(size_t)k * len + i]; s1 += ws[(size_t)(k + SLAB_ROWS) * len + i];
            s2 += ws[(size_t)(k + 2 * SLAB_ROWS) * len + i]; s3 += ws[(size_t)(k + 3 * SLAB_ROWS) * len + i];
        }
        for (; k < nslab; k += SLAB_ROWS) s0 += ws[(size_t)k * len + i];
    }
    red[r][j] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (r == 0 && i < len) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < SLAB_ROWS; ++q) t += red[q][j];
        out[i] = (accumulate ? out[i] : 0.f) + t;
    }
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
               float* ws, float* dw, hipStream_t s, int64_t* ws_bytes_only, int accumulate) {
    WgradParams p; size_t shmem; int threads, grid;
    int rc = plan_own<CB, CBT, KD, KH, KW, S, TPW, OWN_KH>(d, p, shmem, threads, grid);
    if (rc) return rc;
    const int len = CB * d->CA * KD * KH * KW;
    if (ws_bytes_only) { *ws_bytes_only = (int64_t)grid * p.psplit * len * sizeof(float); return VG_OK; }
    vg_launch(wgrad_own_k<CB, CBT, KD, KH, KW, S, TPW, OWN_KH>, dim3(grid), dim3(threads), shmem, s,
              a, b, in_scale, in_shift, ws, p);
    rc = vg_check_launch("wgrad_own");
    if (rc) return rc;
    vg_launch(slab_sum_k, dim3(vg_cdiv(len, 64)), dim3(64 * SLAB_ROWS), 0, s, (const float*)ws, grid * p.psplit, len, accumulate, dw);
    return vg_check_launch("wgrad slab_sum");
}

// ------------------------------------------------------------------------------------------
// MFMA variant: dw[cb][col] (col = (ca, tap)) as a GEMM  D(16 x 16*NT) += A(16 x 4) * B(4 x 16*NT)
// over groups of 4 consecutive positions, with the exact-fp32 matrix instruction
// v_mfma_f32_16x16x4_f32:  A[cb][k] = b[cb][p0+k],  B[k][col] = a[ca][(p0+k)*S + tap - pad].
// A wave keeps ALL of dw (NT accumulator tiles = 4*NT VGPRs) and walks its share of the positions of
// the block's LDS tile: per MFMA one ds_read_b32 (the im2col operand, address = position offset +
// a per-lane column offset computed once) -- no register-level gather, no cross-lane traffic.
// Blocks are persistent over (sample, position-tile) items; slabs are summed by slab_sum_k.
// ------------------------------------------------------------------------------------------
struct WgradMfmaParams {
    vg_wgrad_desc d;
    int TPD, TPH, TPWp;         // position tile (TPWp: PW rounded up to a multiple of 4; the tile spans all of W)
    int tilesH, tilesD;
    int LD, LH, LW, LWp;        // a-tile geometry per channel
    int items;
    int b_off;                  // float offset of the b tile in LDS
    int red_off;                // float offset of the cross-wave reduction buffer (aliases the tiles)
};

template <int NT, int KD, int KH, int KW, int S>
__global__ void __launch_bounds__(256)
wgrad_mfma_k(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ in_scale,
             const float* __restrict__ in_shift, float* __restrict__ ws, WgradMfmaParams p) {
    VG_DYN_SMEM(float, lds);
    constexpr int KVOL = KD * KH * KW;
    constexpr int U = 4;
    const vg_wgrad_desc& d = p.d;
    const int CA = d.CA, CB = d.CB;
    const int ncol = CA * KVOL;
    const int tid = threadIdx.x, lane = tid % VG_WAVE;
    const int wave = vg_wave_id(), nwaves = blockDim.x / VG_WAVE;
    float* atile = lds;
    float* btile = lds + p.b_off;

    // per-lane LDS offset of column (16 t + lane%16) inside the a tile
    int colOff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = t * 16 + (lane & 15);
        int off = 0;
        if (col < ncol) {
            const int ca = col / KVOL, tap = col % KVOL;
            const int kd = tap / (KH * KW), kh = (tap / KW) % KH, kw = tap % KW;
            off = ((ca * p.LD + kd) * p.LH + kh) * p.LWp + kw;
        }
        colOff[t] = off;
    }
    vg_f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t].v[0] = 0.f; acc[t].v[1] = 0.f; acc[t].v[2] = 0.f; acc[t].v[3] = 0.f; }

    const int tiles = p.tilesH * p.tilesD;
    const size_t aplane = (size_t)d.AH * d.AW, bplane = (size_t)d.PH * d.PW;
    const int rl_a = d.pro_on_a ? d.relu_in : 0, rl_b = d.pro_on_a ? 0 : d.relu_in;
    for (int item = blockIdx.x; item < p.items; item += gridDim.x) {
        const int n = item / tiles; const int tile = item % tiles;
        const int thi = tile % p.tilesH, tdi = tile / p.tilesH;
        const int pd0 = tdi * p.TPD, ph0 = thi * p.TPH;
        const int g = (in_scale != nullptr) ? n / d.per_group : 0;
        __syncthreads();
        // ---- a tile [CA][LD][LH][LWp]: prologue applied, zero outside the tensor.  U rows of one (c, dz)
        //      slab per wave and step; the loads are unconditional (clamped addresses) and issued together.
        {
            const int ad0 = pd0 * S - d.pad_d, ah0 = ph0 * S - d.pad_h, aw0 = -d.pad_w;
            for (int c = 0; c < CA; ++c) {
                float sc = 1.f, sh = 0.f;
                if (d.pro_on_a && in_scale) { sc = in_scale[g * CA + c]; sh = in_shift[g * CA + c]; }
                const float* abase = a + ((size_t)n * CA + c) * d.AD * aplane;
                for (int dz = 0; dz < p.LD; ++dz) {
                    const int id = ad0 + dz;
                    const bool dok = id >= 0 && id < d.AD;
                    const float* pbase = abase + (size_t)clampi_(id, 0, d.AD - 1) * aplane;
                    float* dplane = atile + (size_t)(c * p.LD + dz) * p.LH * p.LWp;
                    for (int hy0 = wave * U; hy0 < p.LH; hy0 += nwaves * U) {
                        for (int wx = lane; wx < p.LW; wx += VG_WAVE) {
                            const int iw = aw0 + wx;
                            const bool cok = iw >= 0 && iw < d.AW;
                            const int iwc = clampi_(iw, 0, d.AW - 1);
                            float v[U];
#pragma unroll
                            for (int u = 0; u < U; ++u) v[u] = pbase[(size_t)clampi_(ah0 + hy0 + u, 0, d.AH - 1) * d.AW + iwc];
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                const int hy = hy0 + u, ih = ah0 + hy;
                                if (hy < p.LH)
                                    dplane[hy * p.LWp + wx] = (dok && cok && ih >= 0 && ih < d.AH) ? apply_pro(v[u], rl_a, sc, sh) : 0.f;
                            }
                        }
                    }
                }
            }
        }
        // ---- b tile [TPD][TPH][TPWp][16] (channel fastest, zero for cb >= CB and outside the tensor)
        {
            const int rows = 16 * p.TPD * p.TPH;              // (cb, dz, hy) rows of TPWp positions
            for (int r = wave; r < rows; r += nwaves) {
                const int hy = r % p.TPH; const int t = r / p.TPH; const int dz = t % p.TPD; const int c = t / p.TPD;
                const int pd = pd0 + dz, ph = ph0 + hy;
                const bool ok = c < CB && pd < d.PD && ph < d.PH;
                float sc = 1.f, sh = 0.f;
                if (!d.pro_on_a && in_scale && c < CB) { sc = in_scale[g * CB + c]; sh = in_shift[g * CB + c]; }
                const float* src = b + (((size_t)n * CB + (ok ? c : 0)) * d.PD + (ok ? pd : 0)) * bplane + (size_t)(ok ? ph : 0) * d.PW;
                for (int wx = lane; wx < p.TPWp; wx += VG_WAVE) {
                    const float v = src[min(wx, d.PW - 1)];
                    btile[((size_t)(dz * p.TPH + hy) * p.TPWp + wx) * 16 + c] = (ok && wx < d.PW) ? apply_pro(v, rl_b, sc, sh) : 0.f;
                }
            }
        }
        __syncthreads();
        // ---- position groups of 4 along w
        {
            const int gw = p.TPWp / 4;
            const int groups = p.TPD * p.TPH * gw;
            const int k = lane >> 4;
            for (int gi = wave; gi < groups; gi += nwaves) {
                const int gx = gi % gw; const int t2 = gi / gw; const int py = t2 % p.TPH; const int pz = t2 / p.TPH;
                const int px = gx * 4 + k;
                const float av = btile[((size_t)(pz * p.TPH + py) * p.TPWp + px) * 16 + (lane & 15)];
                const float* ap = atile + (size_t)(pz * S * p.LH + py * S) * p.LWp + px * S;
#pragma unroll
                for (int t = 0; t < NT; ++t) vg_mfma16(av, ap[colOff[t]], acc[t]);
            }
        }
    }
    // ---- cross-wave reduction through LDS (one wave at a time), then one slab per block
    float* red = lds + p.red_off;
    __syncthreads();
    for (int w = 0; w < nwaves; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = (t * 4 + r) * VG_WAVE + lane;
                    red[idx] = (w == 0 ? 0.f : red[idx]) + acc[t].v[r];
                }
        }
        __syncthreads();
    }
    float* out = ws + (size_t)blockIdx.x * CB * ncol;
    for (int i = tid; i < NT * 4 * VG_WAVE; i += blockDim.x) {
        const int l = i % VG_WAVE; const int r = (i / VG_WAVE) % 4; const int t = i / (4 * VG_WAVE);
        const int cb = (l >> 4) * 4 + r, col = t * 16 + (l & 15);
        if (cb < CB && col < ncol) out[(size_t)cb * ncol + col] = red[i];
    }
}

template <int NT, int KD, int KH, int KW, int S>
int launch_mfma(const vg_wgrad_desc* d, const float* a, const float* b, const float* in_scale, const float* in_shift,
                float* ws, float* dw, hipStream_t s, int64_t* ws_bytes_only, int accumulate) {
    WgradMfmaParams p; p.d = *d;
    constexpr int KVOL = KD * KH * KW;
    if (d->CA * KVOL > NT * 16 || d->CB > 16) { vg_set_error("wgrad_mfma: CA=%d CB=%d do not fit NT=%d", d->CA, d->CB, NT); return VG_ERR_UNSUPPORTED; }
    p.TPWp = (d->PW + 3) & ~3;
    p.LW = (p.TPWp - 1) * S + KW; p.LWp = p.LW | 1;
    const size_t budget = 56 * 1024;
    const size_t red_fl = (size_t)NT * 4 * VG_WAVE;
    int best_h = 0, best_d = 0;
    for (int td = 1; td <= 8; ++td)
        for (int th = 1; th <= 16; ++th) {
            if (th > d->PH || td > d->PD) continue;
            const size_t afl = (size_t)d->CA * ((td - 1) * S + KD) * ((th - 1) * S + KH) * p.LWp;
            const size_t bfl = (size_t)td * th * p.TPWp * 16;
            if ((afl + bfl + 8) * 4 <= budget && td * th > best_d * best_h) { best_d = td; best_h = th; }
        }
    if (best_h == 0) { vg_set_error("wgrad_mfma: tile does not fit LDS"); return VG_ERR_UNSUPPORTED; }
    p.TPD = best_d; p.TPH = best_h;
    p.LD = (p.TPD - 1) * S + KD; p.LH = (p.TPH - 1) * S + KH;
    p.tilesH = vg_cdiv(d->PH, p.TPH); p.tilesD = vg_cdiv(d->PD, p.TPD);
    p.items = d->N * p.tilesH * p.tilesD;
    const size_t afl = (size_t)d->CA * p.LD * p.LH * p.LWp;
    p.b_off = (int)((afl + 3) & ~(size_t)3);
    const size_t tile_fl = (size_t)p.b_off + (size_t)p.TPD * p.TPH * p.TPWp * 16;
    p.red_off = 0;
    const size_t shmem = (tile_fl > red_fl ? tile_fl : red_fl) * sizeof(float) + 64;
    int grid = p.items < 1024 ? p.items : 1024;
    const int len = d->CB * d->CA * KVOL;
    if (ws_bytes_only) { *ws_bytes_only = (int64_t)grid * len * sizeof(float); return VG_OK; }
    vg_launch(wgrad_mfma_k<NT, KD, KH, KW, S>, dim3(grid), dim3(256), shmem, s, a, b, in_scale, in_shift, ws, p);
    int rc = vg_check_launch("wgrad_mfma");
    if (rc) return rc;
    vg_launch(slab_sum_k, dim3(vg_cdiv(len, 64)), dim3(64 * SLAB_ROWS), 0, s, (const float*)ws, grid, len, accumulate, dw);
    return vg_check_launch("wgrad slab_sum");
}

// ------------------------------------------------------------------------------------------
// Plane-staged MFMA variant (every layer of the 41x49x35 network).
// Same GEMM as above, but both operands come from LDS images that keep the TENSORS' OWN pitches and are filled
// by flat LDS-DMA copies (global_load_lds_dword, contiguous 256-byte wave-instructions, no VGPR round trip, no
// index arithmetic per element):
//   a slot : the LD whole planes of ONE `a` channel that a block of TPD position planes needs (one contiguous
//            span), double-buffered over the channel loop, which is fully unrolled so that the accumulator tiles
//            of channel ca are named registers;
//   b tile : for each of the (<=16) `b` channels the rows [ph0, ph0+TPH) of the TPD position planes (contiguous).
// Positions are walked as groups of 4 consecutive FLAT (row-major) positions; a lane carries its (dz,py,px)
// incrementally.  ReLU / batch-norm affine are applied to an operand on its way from LDS into the MFMA (VALU work
// beside a 32-cycle matrix instruction); PAD adds the range masks of ConvTranspose3d padding (convt2).
// ------------------------------------------------------------------------------------------
struct WgradPlaneParams {
    vg_wgrad_desc d;
    int TPD, TPH;               // position planes / rows per b tile
    int nph;                    // row blocks per plane block
    int LD;                     // a planes per item
    int a_slot;                 // floats per a buffer (LD*AH*AW + slack, multiple of 64)
    int nbuf;                   // 2: next channel's planes are DMA'd behind the MFMAs; 1: LDS too small for that
    int b_off, bch;             // float offset of the b tile; floats per b channel (== 2 mod 32: conflict-free operand reads)
    int lds_floats;             // total dynamic LDS floats
    int items, pdblocks;
};

template <int CA, int TC, int KD, int KH, int KW, int S, bool PAD>
__global__ void __launch_bounds__(256)
wgrad_plane_k(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ in_scale,
              const float* __restrict__ in_shift, float* __restrict__ ws, WgradPlaneParams p) {
    VG_DYN_SMEM(float, lds);
    constexpr int KVOL = KD * KH * KW;
    constexpr int NT = CA * TC;
    const vg_wgrad_desc& d = p.d;
    const int CB = d.CB;
    const int tid = threadIdx.x, lane = tid % VG_WAVE;
    const int wave = vg_wave_id(), nwaves = blockDim.x / VG_WAVE;
    float* btile = lds + p.b_off;
    const int aplane = d.AH * d.AW, bplane = d.PH * d.PW;

    for (int i = tid; i < p.lds_floats; i += blockDim.x) lds[i] = 0.f;       // slack / unused channels stay finite (zero)

    int colOff[TC];                               // offset of tap (16 t + lane%16) inside an a channel, tensor pitches
    int tkd[TC], tkh[TC], tkw[TC];
#pragma unroll
    for (int t = 0; t < TC; ++t) {
        const int tap = t * 16 + (lane & 15);
        const bool ok = tap < KVOL;
        tkd[t] = ok ? tap / (KH * KW) : 0; tkh[t] = ok ? (tap / KW) % KH : 0; tkw[t] = ok ? tap % KW : 0;
        colOff[t] = (tkd[t] * d.AH + tkh[t]) * d.AW + tkw[t];
    }
    vg_f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t].v[0] = 0.f; acc[t].v[1] = 0.f; acc[t].v[2] = 0.f; acc[t].v[3] = 0.f; }

    const int rl_a = d.pro_on_a ? d.relu_in : 0, rl_b = d.pro_on_a ? 0 : d.relu_in;
    const float lo_a = rl_a ? 0.f : -__builtin_inff(), lo_b = rl_b ? 0.f : -__builtin_inff();
    const int npos = p.TPD * p.TPH * d.PW;                                 // flat positions of one b tile
    const int groups = (npos + 3) / 4;
    const int kq = lane >> 4, cbl = lane & 15;
    // first flat position of this lane and its (dz, py, px); advanced by 4*nwaves per group
    int px0, py0, dz0;
    { const int pf = wave * 4 + kq; px0 = pf % d.PW; const int t2 = pf / d.PW; py0 = t2 % p.TPH; dz0 = t2 / p.TPH; }
    const int step = 4 * nwaves;
    __syncthreads();
    for (int item = blockIdx.x; item < p.items; item += gridDim.x) {
        const int n = item / p.pdblocks; const int pd0 = (item % p.pdblocks) * p.TPD;
        const int g = (in_scale != nullptr) ? n / d.per_group : 0;
        const int ap0 = pd0 * S - d.pad_d;                                // first a plane the tile touches (may be < 0)
        const int pl_lo = max(ap0, 0), pl_hi = min(ap0 + p.LD, d.AD);
        const int nfl = max(pl_hi - pl_lo, 0) * aplane;                   // floats per a channel for this item
        const int adst = (pl_lo - ap0) * aplane;                          // where they land inside the slot
        const float* abase = a + (size_t)n * CA * d.AD * aplane + (size_t)pl_lo * aplane;
        float bsc = 1.f, bsh = 0.f;
        if (!d.pro_on_a && in_scale && cbl < CB) { bsc = in_scale[g * CB + cbl]; bsh = in_shift[g * CB + cbl]; }
        __syncthreads();                                                  // previous item's tiles fully consumed
        // ---- DMA of a channel 0 into buffer 0 (nbuf == CA: the whole item is small enough to keep EVERY channel's
        //      planes resident -- one DMA phase, no barrier inside the channel loop)
        const int nres = (p.nbuf == CA) ? CA : 1;
        for (int c = 0; c < nres; ++c) {
            const float* src = abase + (size_t)c * d.AD * aplane;
            float* dst = lds + c * p.a_slot + adst;
            for (int o = wave * VG_WAVE; o < nfl; o += blockDim.x)
                if (o + lane < nfl) vg_dma4(src + o + lane, dst + o);
        }
        for (int phb = 0; phb < p.nph; ++phb) {
            const int ph0 = phb * p.TPH;
            const int nrow = min(p.TPH, d.PH - ph0);                      // valid rows of this block
            // ---- b tile: per (channel, plane) one contiguous span of rows, flat DMA, tensor pitch
            for (int c = 0; c < CB; ++c)
                for (int dz = 0; dz < p.TPD; ++dz) {
                    if (pd0 + dz >= d.PD) continue;
                    const float* src = b + (((size_t)n * CB + c) * d.PD + pd0 + dz) * bplane + (size_t)ph0 * d.PW;
                    float* dst = btile + c * p.bch + dz * p.TPH * d.PW;
                    const int nb = nrow * d.PW;
                    for (int o = wave * VG_WAVE; o < nb; o += blockDim.x)
                        if (o + lane < nb) vg_dma4(src + o + lane, dst + o);
                }
            vg_dma_wait();
            __syncthreads();
#pragma unroll
            for (int ca = 0; ca < CA; ++ca) {
                float* cur = lds + (ca % p.nbuf) * p.a_slot;
                if (ca + 1 < CA && p.nbuf == 2) {                         // next channel in flight behind this channel's MFMAs
                    float* nxt = lds + ((ca + 1) & 1) * p.a_slot + adst;
                    const float* src = abase + (size_t)(ca + 1) * d.AD * aplane;
                    for (int o = wave * VG_WAVE; o < nfl; o += blockDim.x)
                        if (o + lane < nfl) vg_dma4(src + o + lane, nxt + o);
                }
                float sc = 1.f, sh = 0.f;
                if (d.pro_on_a && in_scale) { sc = in_scale[g * CA + ca]; sh = in_shift[g * CA + ca]; }
                // UG position groups per iteration: all their LDS operand reads are issued before the first MFMA, so the
                // LDS latency (~100+ cycles per dependent read) is paid once per UG*TC matrix instructions, not per one
                constexpr int UG = 4;
                int px = px0, py = py0, dz = dz0;
                const float* bchan = btile + min(cbl, CB - 1) * p.bch;                   // only CB channel slots exist
                const bool cb_ok = cbl < CB;
                for (int gi = wave; gi < groups; gi += nwaves * UG) {
                    float av[UG], bv[UG][TC]; bool pk[UG], okt[UG][TC];
#pragma unroll
                    for (int u = 0; u < UG; ++u) {
                        const bool pok = (gi + u * nwaves < groups) && dz < p.TPD && pd0 + dz < d.PD && py < nrow;
                        pk[u] = pok;
                        const int pf = pok ? (dz * p.TPH + py) * d.PW + px : 0;
                        av[u] = bchan[pf];
                        const int idb = dz * S, ihb = (ph0 + py) * S - d.pad_h, iwb = px * S - d.pad_w;
                        const int aoff = idb * aplane + ihb * d.AW + iwb;
#pragma unroll
                        for (int t = 0; t < TC; ++t) {
                            bool ok = pok;
                            if (PAD) {
                                const int id = ap0 + idb + tkd[t], ih = ihb + tkh[t], iw = iwb + tkw[t];
                                ok = pok && id >= 0 && id < d.AD && ih >= 0 && ih < d.AH && iw >= 0 && iw < d.AW;
                            }
                            okt[u][t] = ok;
                            bv[u][t] = cur[ok ? aoff + colOff[t] : 0];
                        }
                        px += step;
                        while (px >= d.PW) { px -= d.PW; if (++py == p.TPH) { py = 0; ++dz; } }
                    }
#pragma unroll
                    for (int u = 0; u < UG; ++u) {
                        const float a_ = (pk[u] && cb_ok) ? fmaf(fmaxf(av[u], lo_b), bsc, bsh) : 0.f;
#pragma unroll
                        for (int t = 0; t < TC; ++t) {
                            float b_ = bv[u][t];
                            if (d.pro_on_a) b_ = fmaf(fmaxf(b_, lo_a), sc, sh);
                            b_ = okt[u][t] ? b_ : 0.f;
                            vg_mfma16(a_, b_, acc[ca * TC + t]);
                        }
                    }
                }
                if (CA > 1 && p.nbuf != CA) {
                    if (p.nbuf == 2) { vg_dma_wait(); __syncthreads(); }
                    else if (ca + 1 < CA) {                               // single buffer: refill after everyone is done reading
                        __syncthreads();
                        const float* src = abase + (size_t)(ca + 1) * d.AD * aplane;
                        for (int o = wave * VG_WAVE; o < nfl; o += blockDim.x)
                            if (o + lane < nfl) vg_dma4(src + o + lane, lds + adst + o);
                        vg_dma_wait();
                        __syncthreads();
                    }
                }
            }
            if (p.nph > 1 || CA == 1 || p.nbuf == CA) __syncthreads();    // tiles are restaged next
            if (CA > 1 && p.nbuf != CA && p.nph > 1 && phb + 1 < p.nph) { // channel 0 again for the next row block
                for (int o = wave * VG_WAVE; o < nfl; o += blockDim.x)
                    if (o + lane < nfl) vg_dma4(abase + o + lane, lds + adst + o);
            }
        }
    }
    // ---- cross-wave reduction through LDS (one wave at a time), then one slab per block
    float* red = lds;
    __syncthreads();
    for (int w = 0; w < nwaves; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = (t * 4 + r) * VG_WAVE + lane;
                    red[idx] = (w == 0 ? 0.f : red[idx]) + acc[t].v[r];
                }
        }
        __syncthreads();
    }
    const int ncol = CA * KVOL;
    float* out = ws + (size_t)blockIdx.x * CB * ncol;
    for (int i = tid; i < NT * 4 * VG_WAVE; i += blockDim.x) {
        const int l = i % VG_WAVE; const int r = (i / VG_WAVE) % 4; const int t = i / (4 * VG_WAVE);
        const int ca = t / TC, tap = (t % TC) * 16 + (l & 15);
        const int cb = (l >> 4) * 4 + r;
        if (cb < CB && tap < KVOL) out[(size_t)cb * ncol + ca * KVOL + tap] = red[i];
    }
}

// returns -1 when the geometry does not fit (caller falls back)
template <int CA, int TC, int KD, int KH, int KW, int S, bool PAD>
int launch_plane(const vg_wgrad_desc* d, const float* a, const float* b, const float* in_scale, const float* in_shift,
                 float* ws, float* dw, hipStream_t s, int64_t* ws_bytes_only, int accumulate) {
    constexpr int KVOL = KD * KH * KW;
    constexpr int NT = CA * TC;
    const bool padded = d->pad_d || d->pad_h || d->pad_w;
    if (padded != PAD || d->CA != CA || d->CB > 16) return -1;
    if (!PAD && ((d->PD - 1) * S + KD > d->AD || (d->PH - 1) * S + KH > d->AH || (d->PW - 1) * S + KW > d->AW)) return -1;
    WgradPlaneParams p; p.d = *d;
    const int aplane = d->AH * d->AW;
    const size_t budget = 64 * 1024;
    const size_t red_fl = (size_t)NT * 4 * VG_WAVE;
    // slot: LD planes + slack for windows of masked / overhanging positions that run past the last (or before the first) plane
    auto slot_for = [&](int LD) { return (((size_t)(LD + 1) * aplane + (size_t)KH * d->AW + 8 * S + 64 + 63) / 64) * 64; };
    auto bch_for = [&](int td, int th) { size_t f = (size_t)td * th * d->PW + 4; while (f % 32 != 2) ++f; return f; };
    int best_td = 0, best_th = 0, best_nbuf = 0;
    if (CA > 2) {                                   // tiny layers: every channel's planes resident, as many items as possible
        const size_t small_budget = 48 * 1024;
        for (int td = 1; td <= 2 && td <= d->PD && !best_td; ++td)
            if (((size_t)CA * slot_for((td - 1) * S + KD) + d->CB * bch_for(td, d->PH) + 64) * 4 <= small_budget) {
                best_td = td; best_th = d->PH; best_nbuf = CA;
            }
    }
    for (int nbuf = 2; nbuf >= 1 && !best_td; --nbuf) {
        if (nbuf == 2 && CA == 1) continue;
        // smallest tile that fits: LDS per block decides how many blocks (each with its own DMA in flight) share a CU
        // but at least ~128 positions (32 MFMA position groups) per tile, or the barriers outweigh the matrix work
        int td_want = vg_cdiv(128, d->PH * d->PW);
        if (td_want > d->PD) td_want = d->PD;
        for (int td = 1; td <= 8 && td <= d->PD; ++td)
            if ((nbuf * slot_for((td - 1) * S + KD) + d->CB * bch_for(td, d->PH) + 64) * 4 <= budget) {
                best_td = td; best_th = d->PH; best_nbuf = nbuf;
                if (td >= td_want) break;
            }
    }
    if (!best_td)                                   // row blocks of one position plane, single a buffer
        for (int th = d->PH; th >= 4; --th)
            if ((slot_for(KD) + d->CB * bch_for(1, th) + 64) * 4 <= budget) { best_td = 1; best_th = th; best_nbuf = 1; break; }
    if (!best_td) return -1;
    p.TPD = best_td; p.TPH = best_th; p.nbuf = best_nbuf; p.nph = vg_cdiv(d->PH, p.TPH);
    p.LD = (p.TPD - 1) * S + KD;
    p.a_slot = (int)slot_for(p.LD);
    p.bch = (int)bch_for(p.TPD, p.TPH);
    p.b_off = p.nbuf * p.a_slot;
    size_t fl = (size_t)p.b_off + (size_t)d->CB * p.bch + 64;
    if (fl < red_fl) fl = red_fl;
    p.lds_floats = (int)fl;
    p.pdblocks = vg_cdiv(d->PD, p.TPD);
    p.items = d->N * p.pdblocks;
    const int grid = p.items < 1536 ? p.items : 1536;
    const int len = d->CB * CA * KVOL;
    if (ws_bytes_only) { *ws_bytes_only = (int64_t)grid * len * sizeof(float); return VG_OK; }
    vg_launch(wgrad_plane_k<CA, TC, KD, KH, KW, S, PAD>, dim3(grid), dim3(256), fl * sizeof(float), s, a, b, in_scale, in_shift, ws, p);
    int rc = vg_check_launch("wgrad_plane");
    if (rc) return rc;
    vg_launch(slab_sum_k, dim3(vg_cdiv(len, 64)), dim3(64 * SLAB_ROWS), 0, s, (const float*)ws, grid, len, accumulate, dw);
    return vg_check_launch("wgrad slab_sum");
}

// ------------------------------------------------------------------------------------------
// Row-walking MFMA variant (layers whose position rows are >= 12 wide: the four large decoder layers, conv1-3).
// Same GEMM and operand images as wgrad_plane_k, two changes that matter:
//  * a WAVE owns whole position rows, so (plane, row) are scalars and a k-step (4 consecutive positions of the row)
//    addresses both operands as  row base + immediate : ~8 VALU instructions beside TC matrix instructions instead of
//    the ~35 of the flat-position walk (div/mod-free carry of (dz,py,px) per lane, masks, offset rebuilds);
//  * the `a` slot holds only the ROWS the tile's windows touch ((TPH-1)*S+KH rows of each plane, still one contiguous
//    span per plane for the flat LDS-DMA), so a tile is ~20-30 KB instead of 60 KB and 3-8 blocks share a CU: one
//    block's DMA + barriers hide behind the others' MFMAs.
// ------------------------------------------------------------------------------------------
#ifdef VG_EMU
#define VG_WG_FENCE() ((void)0)
#else
#define VG_WG_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef VG_WG_MINB
#define VG_WG_MINB 1
#endif
#ifdef VG_STAMP
// Diagnostic build only (tools/diag): per-wave cycle sums of wgrad_rows_k's phases, read back through vg_stamp_read_wg.
__device__ unsigned long long vg_wg_stamp_out[2048 * 4 * 8];
#define VG_WS_T(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define VG_WS_ADD(i) do { unsigned long long t_; VG_WS_T(t_); st_sum[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define VG_WS_ADD(i) do {} while (0)
#endif
template <int V> struct vg_int { static constexpr int value = V; };
__host__ __device__ constexpr int vg_ival(int v) { return v; }
template <int V> __host__ __device__ constexpr int vg_ival(vg_int<V>) { return V; }
template <int N, int I = 0, typename F>
__host__ __device__ inline void vg_static_for(F&& f) { if constexpr (I < N) { f(vg_int<I>{}); vg_static_for<N, I + 1>(f); } }

struct WgradRowsParams {
    vg_wgrad_desc d;
    int TPD, TPH, nph, pdblocks;
    int LD, AR, apl;            // a planes / rows per plane in a slot; floats per slot plane (AR*AW)
    int a_slot, a_front;        // floats per a channel slot (with slack both ends); front slack
    int nbuf;                   // CA: all channels resident; 2: double-buffered over the channel loop; 1: single
    int b_off, bch;
    int lds_floats;
    int items;
    int wave_slabs;             // small grids: every wave writes its own slab (no cross-wave LDS reduction: 4 serial rounds, ~13 us)
    // grouped mode (vg_wgrad3d_grouped, template GRP): the grid is split evenly over the batch-norm groups (sample / per_group)
    int grp_items;              // items per group
    int ipb;                    // blocks per group
    int ones_row;               // 1: MFMA row CB carries a constant-one position channel -> per-tap sums of the window tensor
};

// DSH (stride-2 layers with 8 position channels, KD >= 2*S): the 8 idle MFMA rows carry the SAME channels one position plane further on.
// With the window operand restricted to the taps kd' in [S, KD), row (h = 0, cb) accumulates dw[cb][.][kd'] and row (h = 1, cb) -- whose
// position is one plane = S window planes ahead -- accumulates dw[cb][.][kd' - S]: together every kd in [0, KD), from (KD-S)*KH*KW instead
// of KD*KH*KW window taps (convt4: 27 instead of 45 = 2 instead of 3 matrix instructions per k-step and channel; the 4x4x4 layer of the
// 82x98x70 geometry: 32 instead of 64 = 2 instead of 4).  The position planes of a tile start at -1 (row h = 1 covers plane 0 there).
// ONE != 0: a row is ONE >> 1 unrolled blocks of UG k-steps at COMPILE-TIME offsets (rows of 13..16 positions = one block of 4: every
// large layer of the 41x49x35 network; 33 positions = three blocks of 3: its first and last layer): the LDS reads then carry their
// offsets as immediates (with the run-time step loop each read had its own address add: 39 vector + 23 scalar instructions per row
// beside 8 MFMAs, ISA of the convt4 instance); ONE & 1 (PW a multiple of 4): no position mask in the last block either.
// ONE == 0: the general step loop.  (convt4's weight gradient 627 -> 507 us, convt3's 361 -> 313.)
template <int CA, int TC, int KD, int KH, int KW, int S, bool PAD, bool PA, int UG, bool RES, bool GRP = false, bool DSH = false, int ONE = 0>
__global__ void __launch_bounds__(256, (CA * TC >= 32 ? 2 : CA * TC >= 16 ? VG_WG_MINB : 1))         // 32 accumulator tiles: keep two waves per SIMD (<= 256 registers)
wgrad_rows_k(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ in_scale,
             const float* __restrict__ in_shift, float* __restrict__ ws, WgradRowsParams p) {
    VG_DYN_SMEM(float, lds);
    constexpr int KVOL = KD * KH * KW;
    constexpr int KVW = DSH ? (KD - S) * KH * KW : KVOL;               // window taps the matrix columns enumerate
    constexpr int NT = CA * TC;
    const vg_wgrad_desc& d = p.d;
    const int CB = d.CB;
    const int tid = threadIdx.x, lane = tid % VG_WAVE;
    const int wave = vg_wave_id(), nwaves = blockDim.x / VG_WAVE;
    float* btile = lds + p.b_off;
    const int kq = lane >> 4, cbl = lane & 15;

    for (int i = tid; i < p.lds_floats; i += blockDim.x) lds[i] = 0.f;       // never-written words stay finite

    int colOff[TC], tkd[TC], tkh[TC], tkw[TC];
#pragma unroll
    for (int t = 0; t < TC; ++t) {
        const int tap = t * 16 + cbl;
        const bool ok = tap < KVW;
        tkd[t] = ok ? tap / (KH * KW) : 0; tkh[t] = ok ? (tap / KW) % KH : 0; tkw[t] = ok ? tap % KW : 0;     // DSH: tkd counts from kd' = S
        colOff[t] = tkd[t] * p.apl + tkh[t] * d.AW + tkw[t];
    }
    vg_f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t].v[0] = 0.f; acc[t].v[1] = 0.f; acc[t].v[2] = 0.f; acc[t].v[3] = 0.f; }

    const int rl_a = PA ? d.relu_in : 0, rl_b = PA ? 0 : d.relu_in;           // PA: the ReLU / BN affine belongs to the window tensor
    const float lo_a = rl_a ? 0.f : -__builtin_inff(), lo_b = rl_b ? 0.f : -__builtin_inff();
    const int ksteps = (d.PW + 3) / 4;
    const int aplane = d.AH * d.AW, bplane = d.PH * d.PW;
    const bool cb_ok = cbl < (DSH ? 2 * CB : CB);
    // DSH: lanes CB .. 2CB-1 read the same channels one position plane (TPH*PW floats of the tile) further on
    const float* bchan = btile + (DSH ? (cbl % CB) : min(cbl, CB - 1)) * p.bch + ((DSH && cbl >= CB) ? p.TPH * d.PW : 0) + kq;
    __syncthreads();

    const int CBW = CB + (GRP ? 1 : 0);                                  // rows of dw written out (grouped: + the ones row)
    const int ncol_o = CA * KVOL;
    // matrix row cbr, window tap -> (row of dw, tap of dw), or -1: plain = (cbr, tap); DSH: see above
    auto out_index = [&](int cbr, int tap, int ca) -> long long {
        if (!DSH) return (cbr < CBW && tap < KVOL) ? (long long)cbr * ncol_o + ca * KVOL + tap : -1;
        if (cbr >= 2 * CB || tap >= KVW) return -1;
        const int h = cbr / CB, cb = cbr % CB, kdw = tap / (KH * KW), rest = tap % (KH * KW);
        if (h == 1 && kdw >= S) return -1;                                // a duplicate of row h = 0's kd = kdw
        return (long long)cb * ncol_o + ca * KVOL + (h == 0 ? kdw + S : kdw) * (KH * KW) + rest;
    };
    // ---- close a slab: the block's (or, small grids, each wave's) partial dw -> workspace
    auto write_out = [&](int slab) {
        if (p.wave_slabs) {
            float* out = ws + ((size_t)slab * nwaves + wave) * CBW * ncol_o;
            const int cb0 = (lane >> 4) * 4, tapl = lane & 15;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int ca = t / TC, tap = (t % TC) * 16 + tapl;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long oi = out_index(cb0 + r, tap, ca);
                    if (oi >= 0) out[oi] = acc[t].v[r];
                }
                VG_WG_FENCE();                      // one tile at a time: otherwise all NT*4 accumulators are copied to VGPRs up front
            }
            return;
        }
        // cross-wave reduction through LDS (one wave at a time), then one slab
        float* red = lds;
        __syncthreads();
        for (int w = 0; w < nwaves; ++w) {
            if (wave == w) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = (t * 4 + r) * VG_WAVE + lane;
                        red[idx] = (w == 0 ? 0.f : red[idx]) + acc[t].v[r];
                        if (r == 3) VG_WG_FENCE();
                    }
            }
            __syncthreads();
        }
        float* out = ws + (size_t)slab * CBW * ncol_o;
        for (int i = tid; i < NT * 4 * VG_WAVE; i += blockDim.x) {
            const int l = i % VG_WAVE; const int r = (i / VG_WAVE) % 4; const int t = i / (4 * VG_WAVE);
            const int ca = t / TC, tap = (t % TC) * 16 + (l & 15);
            const int cb = (l >> 4) * 4 + r;
            const long long oi = out_index(cb, tap, ca);
            if (oi >= 0) out[oi] = red[i];
        }
    };
    // plain mode: items blockIdx.x, +gridDim.x, ...   grouped mode: the grid is split evenly over the batch-norm groups (ipb = blocks per
    // group); a group's blocks deal ITS items round-robin, so every slab belongs to one group and blocks running side by side still
    // work on neighbouring tiles (their halos meet in L2; contiguous item ranges per block cost 0.2 ms in re-fetched planes)
    const int it_step = GRP ? p.ipb : (int)gridDim.x;
    const int it0 = GRP ? ((int)blockIdx.x / p.ipb) * p.grp_items + (int)blockIdx.x % p.ipb : (int)blockIdx.x;
    const int it1 = GRP ? ((int)blockIdx.x / p.ipb + 1) * p.grp_items : p.items;

#ifdef VG_STAMP
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last;
    VG_WS_T(st_last);
#endif
    for (int item = it0; item < it1; item += it_step) {
        const int n = item / (p.pdblocks * p.nph); const int rem = item % (p.pdblocks * p.nph);
        const int pd0 = (rem / p.nph) * p.TPD - (DSH ? 1 : 0), ph0 = (rem % p.nph) * p.TPH;
        const int nrow = min(p.TPH, d.PH - ph0), ndz = min(p.TPD, d.PD - pd0);
        const int g = (in_scale != nullptr) ? n / d.per_group : 0;
        const int ap0 = pd0 * S - d.pad_d + (DSH ? S : 0), ih0 = ph0 * S - d.pad_h;
        const int pl_lo = max(ap0, 0), pl_hi = min(ap0 + p.LD, d.AD);
        const int r_lo = max(ih0, 0), r_hi = min(ih0 + p.AR, d.AH);
        const int npl = max(pl_hi - pl_lo, 0);
        const int cnt = max(r_hi - r_lo, 0) * d.AW;                         // contiguous floats per staged plane
        const int adst = p.a_front + (pl_lo - ap0) * p.apl + (r_lo - ih0) * d.AW;
        const float* abase = a + (((size_t)n * CA * d.AD + pl_lo) * d.AH + r_lo) * d.AW;
        float bsc = cb_ok ? 1.f : 0.f, bsh = 0.f;                        // lanes without a b channel contribute zeros
        if (!PA && in_scale && cb_ok) { bsc = in_scale[g * CB + (DSH ? cbl % CB : cbl)]; bsh = in_shift[g * CB + (DSH ? cbl % CB : cbl)]; }
        if (GRP && cbl == CB) bsh = 1.f;                                 // ones row (bsc stays 0: the lane reads channel CB-1's finite data)
        VG_WS_ADD(0);                                                     // item set-up
        __syncthreads();                                                  // previous item's tiles fully consumed
        VG_WS_ADD(1);                                                     // barrier (item start)
        // one (channel, plane) span per wave at a time: the span's base addresses are formed once, the 256-byte DMA
        // instructions of the span then cost a handful of scalar adds each (a flattened loop pays two scalar divisions
        // and 64-bit address arithmetic -- ~75 SALU instructions -- per DMA instruction)
        auto stage_a = [&](int c0, int nc, int slot0) {
            for (int c = 0; c < nc; ++c) {
                const float* src = abase + ((size_t)(c0 + c) * d.AD + wave) * aplane;
                float* dst = lds + (slot0 + c) * p.a_slot + adst + wave * p.apl;
                for (int pl = wave; pl < npl; pl += 4, src += 4 * (size_t)aplane, dst += 4 * p.apl) {
                    // (the 16-byte form, vg_dma_block, measured here: convt4 802 -> 793 us, convt5 508 -> 490, but convt2 224 -> 250-303 and the
                    //  small layers 5-20 % slower -- spans of a few hundred floats: its head / tail pieces cost what the wide body saves)
                    vg_dma_span(src + lane, dst, cnt, lane);
                }
            }
        };
        if (RES) stage_a(0, CA, 0);
        else stage_a(0, 1, 0);
        {
            const int nb = nrow * d.PW;
            const size_t bch_g = (size_t)d.PD * bplane;
            const float* src_c = b + ((size_t)n * CB + wave) * bch_g + (long long)pd0 * bplane + (size_t)ph0 * d.PW;     // (DSH: pd0 may be -1; such planes are not read)
            float* dst_c = btile + wave * p.bch;
            const int ndzs = DSH ? ndz + 1 : ndz;                         // DSH: + the plane behind the tile (rows h = 1 of its last plane)
            for (int c = wave; c < CB; c += 4, src_c += 4 * bch_g, dst_c += 4 * p.bch) {
                const float* src = src_c; float* dst = dst_c;
                for (int dz = 0; dz < ndzs; ++dz, src += bplane, dst += p.TPH * d.PW) {
                    if (!DSH || (pd0 + dz >= 0 && pd0 + dz < d.PD)) vg_dma_span(src + lane, dst, nb, lane);
                    else for (int o = lane; o < nb; o += VG_WAVE) dst[o] = 0.f;      // a plane outside the tensor: its positions contribute nothing
                }
            }
        }
        VG_WS_ADD(2);                                                     // copy issue (a slot 0 + b tile)
        vg_dma_wait();
        VG_WS_ADD(3);                                                     // copy wait
        __syncthreads();
        VG_WS_ADD(1);
        // DSH: a position plane outside the tensor (plane -1 of rows h = 0, plane PD of rows h = 1) contributes exactly nothing, whatever
        // the prologue's shift: its lanes' scale and shift are zeroed per plane (the tile holds zeros there)
        float bscz = bsc, bshz = bsh;
        auto plane_factors = [&](int dz) {
            if (DSH) {
                const int pl = pd0 + dz + (cbl >= CB ? 1 : 0);
                const bool v = pl >= 0 && pl < d.PD;
                bscz = v ? bsc : 0.f; bshz = v ? bsh : 0.f;
            }
        };
        // one position row against one `a` channel: UG k-steps per iteration, all operand reads first, then the matrix
        // instructions.  ONE branch-free loop whose trip count is rounded up to UG (surplus k-steps have px >= PW: their A
        // operand is zeroed, their reads stay inside the tiles' slack); only the LAST block can hold such positions and it
        // alone carries the mask.  (A remainder branch would split the accumulators' live ranges: the compiler then
        // shuffles all NT*4 of them between register sets on every row.)
        auto row_channel = [&](auto ca_tag, const float* bp, const float* ap, float sc, float sh, const bool* okdh) {
            constexpr int ca = decltype(ca_tag)::value;
            auto kblock = [&](auto masked_tag, auto ks_) {         // ks_: int, or vg_int<0> (ONE: compile-time 0)
                constexpr bool MASKED = decltype(masked_tag)::value != 0;
                const int ks = vg_ival(ks_);
                float av[UG], bv[UG][TC];
#pragma unroll
                for (int u = 0; u < UG; ++u) {
                    av[u] = bp[(ks + u) * 4];
#pragma unroll
                    for (int t = 0; t < TC; ++t) bv[u][t] = ap[colOff[t] + (ks + u) * 4 * S];
                }
#pragma unroll
                for (int u = 0; u < UG; ++u) {
                    const int px = (ks + u) * 4 + kq;
                    float a_ = fmaf(vg_max(av[u], lo_b), bscz, bshz);             // PA: lo_b = -inf, bsc = 1 (0 for idle lanes), bsh = 0
                    if (MASKED) a_ = px < d.PW ? a_ : 0.f;
#pragma unroll
                    for (int t = 0; t < TC; ++t) {
                        float b_ = bv[u][t];
                        if (PA) b_ = fmaf(vg_max(b_, lo_a), sc, sh);
                        if (PAD) {
                            const int iw = px * S - d.pad_w + tkw[t];
                            b_ = (okdh[t] && iw >= 0 && iw < d.AW) ? b_ : 0.f;
                        }
                        vg_mfma16(a_, b_, acc[ca * TC + t]);
                    }
                }
            };
            if constexpr (ONE != 0) {
                constexpr int NB = ONE >> 1;
                vg_static_for<NB>([&](auto i_tag) {
                    constexpr int I = decltype(i_tag)::value;
                    if constexpr (I + 1 < NB || (ONE & 1)) kblock(vg_int<0>{}, vg_int<I * UG>{});
                    else kblock(vg_int<1>{}, vg_int<I * UG>{});
                });
            } else {
                int ks = 0;
                for (; ks < ksteps - UG; ks += UG) kblock(vg_int<0>{}, ks);
                kblock(vg_int<1>{}, ks);
            }
        };
        auto row_masks = [&](int dz, int py, bool* okdh) {
#pragma unroll
            for (int t = 0; t < TC; ++t) {
                const int id = ap0 + dz * S + tkd[t], ih = ih0 + py * S + tkh[t];
                okdh[t] = !PAD || (id >= 0 && id < d.AD && ih >= 0 && ih < d.AH);
            }
        };
        if constexpr (RES) {
            // every channel's window rows are resident: rows outermost, so the row bookkeeping (and the masks) are paid
            // once per row, not once per (row, channel) -- the small layers (5..7 positions per row) live on this
            float scv[CA], shv[CA];
#pragma unroll
            for (int ca = 0; ca < CA; ++ca) {
                scv[ca] = 1.f; shv[ca] = 0.f;
                if (PA && in_scale) { scv[ca] = in_scale[g * CA + ca]; shv[ca] = in_shift[g * CA + ca]; }
            }
            for (int dz = 0; dz < ndz; ++dz)
            for (int py = (wave - dz * nrow) & 3; py < nrow; py += 4) {
                plane_factors(dz);
                const float* bp = bchan + (dz * p.TPH + py) * d.PW;
                const float* ap0_ = lds + p.a_front + (dz * S) * p.apl + (py * S) * d.AW + kq * S - d.pad_w;
                bool okdh[TC];
                row_masks(dz, py, okdh);
                vg_static_for<CA>([&](auto ca_tag) {
                    constexpr int ca = decltype(ca_tag)::value;
                    row_channel(ca_tag, bp, ap0_ + ca * p.a_slot, scv[ca], shv[ca], okdh);
                });
            }
        } else {
            vg_static_for<CA>([&](auto ca_tag) {
                constexpr int ca = decltype(ca_tag)::value;
                const float* cur = lds + (ca % p.nbuf) * p.a_slot;
                if (ca + 1 < CA && p.nbuf == 2) stage_a(ca + 1, 1, (ca + 1) & 1);    // in flight behind this channel's MFMAs
                VG_WS_ADD(4);                                                     // copy issue (next channel)
                float sc = 1.f, sh = 0.f;
                if (PA && in_scale) { sc = in_scale[g * CA + ca]; sh = in_shift[g * CA + ca]; }
                // rows of the tile dealt round-robin over the 4 waves ((plane, row) wave-uniform: scalars, no division)
                for (int dz = 0; dz < ndz; ++dz)
                for (int py = (wave - dz * nrow) & 3; py < nrow; py += 4) {
                    plane_factors(dz);
                    const float* bp = bchan + (dz * p.TPH + py) * d.PW;
                    const float* ap = cur + p.a_front + (dz * S) * p.apl + (py * S) * d.AW + kq * S - d.pad_w;
                    bool okdh[TC];
                    row_masks(dz, py, okdh);
                    row_channel(ca_tag, bp, ap, sc, sh, okdh);
                }
                VG_WS_ADD(5);                                                     // matrix work of a channel
                if (CA > 1) {
                    if (p.nbuf == 2) { vg_dma_wait(); VG_WS_ADD(6); __syncthreads(); VG_WS_ADD(7); }
                    else if (ca + 1 < CA) { __syncthreads(); stage_a(ca + 1, 1, 0); vg_dma_wait(); __syncthreads(); }
                }
            });
        }
    }
#ifdef VG_STAMP
    if (lane == 0 && blockIdx.x < 2048)
        for (int i = 0; i < 8; ++i) vg_wg_stamp_out[(blockIdx.x * 4 + wave) * 8 + i] = st_sum[i];
#endif
    write_out((int)blockIdx.x);
}

#ifdef VG_STAMP
extern "C" int vg_stamp_read_wg(unsigned long long* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(vg_wg_stamp_out), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
#endif

// Slabs of a grouped launch -> out[g][len]: group g owns the slabs [g*spg, (g+1)*spg)
__global__ void __launch_bounds__(64 * SLAB_ROWS)
slab_sum_groups_k(const float* __restrict__ ws, int spg, int len, float* __restrict__ out) {
    __shared__ float red[SLAB_ROWS][64];
    const int j = threadIdx.x % 64, r = threadIdx.x / 64;
    const int i = blockIdx.x * 64 + j, g = blockIdx.y;
    float s0 = 0.f;
    if (i < len)
        for (int k = g * spg + r; k < (g + 1) * spg; k += SLAB_ROWS) s0 += ws[(size_t)k * len + i];
    red[r][j] = s0;
    __syncthreads();
    if (r == 0 && i < len) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < SLAB_ROWS; ++q) t += red[q][j];
        out[(size_t)g * len + i] = t;
    }
}

// returns -1 when the geometry does not fit (caller falls back)
template <int CA, int TC, int KD, int KH, int KW, int S, bool PAD, bool DSH = false>
int launch_rows(const vg_wgrad_desc* d, const float* a, const float* b, const float* in_scale, const float* in_shift,
                float* ws, float* dw, hipStream_t s, int64_t* ws_bytes_only, int accumulate, int grouped = 0) {
    constexpr int KVOL = KD * KH * KW;
    constexpr int NT = CA * TC;
    constexpr int KDW = DSH ? KD - S : KD;                              // window planes a position touches (DSH: taps kd' in [S, KD) only)
    const bool padded = d->pad_d || d->pad_h || d->pad_w;
    if (padded != PAD || d->CA != CA || d->CB > 16) return -1;
    if (DSH && (2 * d->CB > 16 || grouped || KD < 2 * S)) return -1;       // (never on in_scale: the workspace query passes none)
    const int PDE = DSH ? d->PD + 1 : d->PD;                            // position planes the tiles cover (DSH: from -1)
    const bool narrow = d->PW < 12;                 // 5..7-position rows: only worth it with every channel resident (rows outermost)
    if (!PAD && ((d->PD - 1) * S + KD > d->AD || (d->PH - 1) * S + KH > d->AH || (d->PW - 1) * S + KW > d->AW)) return -1;
    // LDS per block: measured sweep (tools/layer_bench.py over a build-time cap): 56 KB is best or equal for every layer of the net
    // (convt4 212 -> 187 us, convt2 104 -> 92, convt5 130 -> 119 vs 24-48 KB); 64 KB leaves one block per CU too few
    // (measured and NOT adopted: compiling the 16-24-tile instances for three waves per SIMD (-DVG_WG_MINB=3, 166 registers, no spills) with
    //  48 KB tiles is 6 % faster per layer in tools/layer_bench.py -- convt4 799 -> 753 us, convt3 345 -> 326 -- but not in the step, where these
    //  launches share the GPU with the gain block's backward on the second stream: 8.17-8.22 vs 8.21-8.24 ms)
    const size_t cap = (size_t)56 * 1024;
    (void)narrow;
    const size_t red_fl = (size_t)NT * 4 * VG_WAVE;
    const int front = 4;                            // >= pad_w: the first window of a padded row starts before the slot's row
    WgradRowsParams best; double best_score = -1;
    for (int td = 1; td <= 4 && td <= d->PD; ++td)
        for (int th = 1; th <= d->PH; ++th) {
            WgradRowsParams p; p.d = *d;
            p.TPD = td; p.TPH = th; p.LD = (td - 1) * S + KDW; p.AR = (th - 1) * S + KH; p.apl = p.AR * d->AW;
            p.a_front = front;
            p.a_slot = (int)((((size_t)p.LD * p.apl + front + 4 + 16 * S + 3) / 4) * 4);     // back slack: rounded-up k-steps of the last row
            size_t f = (size_t)(td + (DSH ? 1 : 0)) * th * d->PW + 24; while (f % 32 != 2) ++f;       // +24: rounded-up k-steps read past the last row
            p.bch = (int)f;
            p.nbuf = 0;
            const int opts[3] = {CA, 2, 1};
            for (int k = 0; k < 3 && !p.nbuf; ++k) {
                const int nb = opts[k];
                if (nb > CA || (nb == 2 && CA <= 2 && k == 1)) continue;
                if (((size_t)nb * p.a_slot + (size_t)d->CB * p.bch + 64) * 4 <= cap) p.nbuf = nb;
            }
            if (!p.nbuf || (narrow && p.nbuf != CA)) continue;
            const int rows = td * th, pos = rows * d->PW;
            const double util = ((double)d->PH / (vg_cdiv(d->PH, th) * th)) * ((double)PDE / (vg_cdiv(PDE, td) * td)) *
                                ((double)rows / (4 * vg_cdiv(rows, 4)));
            const double halo = (double)(td * S) * (th * S) / ((double)p.LD * p.AR);
            double score = util * pos / (pos + 96.0) * (0.6 + 0.4 * halo);
            if (p.nbuf == 1 && CA > 1) score *= 0.6;
            // few-item layers (encoder end, 32 samples): one wave per SIMD runs its whole dependent chain exposed -- prefer
            // tiles small enough that every CU gets a couple of blocks
            const long items = (long)d->N * vg_cdiv(PDE, td) * vg_cdiv(d->PH, th);
            if (items < 512) score *= ((double)items / 512.0) * ((double)items / 512.0);
            if (score > best_score) { best_score = score; best = p; }
        }
    if (best_score < 0) return -1;
    WgradRowsParams p = best;
    p.nph = vg_cdiv(d->PH, p.TPH); p.pdblocks = vg_cdiv(PDE, p.TPD);
    p.b_off = p.nbuf * p.a_slot;
    size_t fl = (size_t)p.b_off + (size_t)d->CB * p.bch + 64;
    if (fl < red_fl) fl = red_fl;
    p.lds_floats = (int)fl;
    p.items = d->N * p.pdblocks * p.nph;
    // k-steps (4 positions) per row, rounded up to the unroll that wastes the fewest
    const int ksteps = vg_cdiv(d->PW, 4);
    int ug = 4, waste = vg_cdiv(ksteps, 4) * 4 - ksteps;
    for (int u = 3; u >= 2; --u) { const int w_ = vg_cdiv(ksteps, u) * u - ksteps; if (w_ < waste) { waste = w_; ug = u; } }
    using kern_t = void (*)(const float*, const float*, const float*, const float*, float*, WgradRowsParams);
    kern_t kern;
    const bool res = CA > 1 && p.nbuf == CA;
#define VG_PICK(PA_, RES_) (ug == 4 ? wgrad_rows_k<CA, TC, KD, KH, KW, S, PAD, PA_, 4, RES_, false, DSH> : ug == 3 ? wgrad_rows_k<CA, TC, KD, KH, KW, S, PAD, PA_, 3, RES_, false, DSH> : wgrad_rows_k<CA, TC, KD, KH, KW, S, PAD, PA_, 2, RES_, false, DSH>)
    if (CA > 1 && res) kern = d->pro_on_a ? VG_PICK(true, true) : VG_PICK(false, true);
    else kern = d->pro_on_a ? VG_PICK(true, false) : VG_PICK(false, false);
#undef VG_PICK
    if (!res && ug == 4 && ksteps == 4) {                   // one block per row (see ONE): convt4 / convt3 / conv2 / conv3 at 41x49x35
        if (d->PW % 4 == 0) kern = d->pro_on_a ? wgrad_rows_k<CA, TC, KD, KH, KW, S, PAD, true, 4, false, false, DSH, 3> : wgrad_rows_k<CA, TC, KD, KH, KW, S, PAD, false, 4, false, false, DSH, 3>;
        else kern = d->pro_on_a ? wgrad_rows_k<CA, TC, KD, KH, KW, S, PAD, true, 4, false, false, DSH, 2> : wgrad_rows_k<CA, TC, KD, KH, KW, S, PAD, false, 4, false, false, DSH, 2>;
    }
    if constexpr (CA > 1 && !DSH) {                         // ... and the narrow rows (4..8 positions, every channel resident): convt1 / convt2 / conv4 / conv5
        if (res && ug == 2 && ksteps <= 2) {
            if (d->PW == 8) kern = d->pro_on_a ? wgrad_rows_k<CA, TC, KD, KH, KW, S, PAD, true, 2, true, false, false, 3> : wgrad_rows_k<CA, TC, KD, KH, KW, S, PAD, false, 2, true, false, false, 3>;
            else kern = d->pro_on_a ? wgrad_rows_k<CA, TC, KD, KH, KW, S, PAD, true, 2, true, false, false, 2> : wgrad_rows_k<CA, TC, KD, KH, KW, S, PAD, false, 2, true, false, false, 2>;
        }
    }
    if constexpr (CA == 1 && !DSH) {                        // three blocks of 3 per row: the 33-position rows of conv1 / convt5
        if (!res && ug == 3 && ksteps == 9 && d->PW % 4 != 0)
            kern = d->pro_on_a ? wgrad_rows_k<CA, TC, KD, KH, KW, S, PAD, true, 3, false, false, false, 6> : wgrad_rows_k<CA, TC, KD, KH, KW, S, PAD, false, 3, false, false, false, 6>;
    }
    if (grouped) {
        if constexpr (CA == 1 && !PAD) {
            if (d->pro_on_a) return -1;
            kern = ug == 4 ? wgrad_rows_k<CA, TC, KD, KH, KW, S, PAD, false, 4, false, true> : ug == 3 ? wgrad_rows_k<CA, TC, KD, KH, KW, S, PAD, false, 3, false, true>
                           : wgrad_rows_k<CA, TC, KD, KH, KW, S, PAD, false, 2, false, true>;
            if (ug == 3 && ksteps == 9 && d->PW % 4 != 0) kern = wgrad_rows_k<CA, TC, KD, KH, KW, S, PAD, false, 3, false, true, false, 6>;
        } else return -1;
    }
    int per_cu = vg_blocks_per_cu((const void*)kern, 256, fl * sizeof(float));   // persistent grid == resident blocks
    if (per_cu > 8) per_cu = 8;
    int grid = 256 * per_cu; if (grid > p.items) grid = p.items;
    p.grp_items = 0; p.ipb = 0; p.ones_row = 0;
    if (grouped) {
        // per-group partials (+ the ones row): the same number of blocks for every group
        if (d->CB >= 16) return -1;                                    // the ones row needs a free MFMA row
        const int G = d->N / d->per_group;
        p.grp_items = d->per_group * p.pdblocks * p.nph; p.ones_row = 1;
        p.ipb = grid / G; if (p.ipb < 1) p.ipb = 1;
        if (p.ipb > p.grp_items) p.ipb = p.grp_items;
        grid = p.ipb * G;
    }
    const int len = (d->CB + p.ones_row) * CA * KVOL;
    p.wave_slabs = grid <= 512 ? 1 : 0;
    const int per_slab = p.wave_slabs ? 4 : 1;
    const int nslabs = grid * per_slab;
    if (ws_bytes_only) { *ws_bytes_only = (int64_t)nslabs * len * sizeof(float); return VG_OK; }
    vg_launch(kern, dim3(grid), dim3(256), fl * sizeof(float), s, a, b, in_scale, in_shift, ws, p);
    int rc = vg_check_launch("wgrad_rows");
    if (rc) return rc;
    if (grouped) {
        const int G = d->N / d->per_group;
        vg_launch(slab_sum_groups_k, dim3(vg_cdiv(len, 64), G), dim3(64 * SLAB_ROWS), 0, s, (const float*)ws, p.ipb * per_slab, len, dw);
        return vg_check_launch("wgrad slab_sum_groups");
    }
    vg_launch(slab_sum_k, dim3(vg_cdiv(len, 64)), dim3(64 * SLAB_ROWS), 0, s, (const float*)ws, nslabs, len, accumulate, dw);
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
                float* ws, float* dw, hipStream_t s, int64_t* ws_bytes_only, int accumulate) {
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
    vg_launch(slab_sum_k, dim3(vg_cdiv(len, 64)), dim3(64 * SLAB_ROWS), 0, s, (const float*)ws, grid, len, accumulate, dw);
    return vg_check_launch("wgrad slab_sum");
}

int dispatch(const vg_wgrad_desc* d, const float* a, const float* b, const float* in_scale, const float* in_shift,
             float* ws, float* dw, hipStream_t s, int64_t* ws_only, int accumulate) {
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
#define PLANE(CA, TC, KD, KH, KW, S) \
    { { int r_ = launch_rows<CA, TC, KD, KH, KW, S, false>(d, a, b, in_scale, in_shift, ws, dw, s, ws_only, accumulate); if (r_ >= 0) return r_; } \
      int r_ = launch_plane<CA, TC, KD, KH, KW, S, false>(d, a, b, in_scale, in_shift, ws, dw, s, ws_only, accumulate); if (r_ >= 0) return r_; }
    if (d->CB <= 16 && d->PW <= 128) {
        if (k333 && d->CA == 1 && d->stride == 1) PLANE(1, 2, 3, 3, 3, 1);
        if (k333 && d->CA == 16 && d->stride == 2 && (d->pad_d || d->pad_h || d->pad_w))
            { { int r_ = launch_rows<16, 2, 3, 3, 3, 2, true>(d, a, b, in_scale, in_shift, ws, dw, s, ws_only, accumulate); if (r_ >= 0) return r_; }
              int r_ = launch_plane<16, 2, 3, 3, 3, 2, true>(d, a, b, in_scale, in_shift, ws, dw, s, ws_only, accumulate); if (r_ >= 0) return r_; }
        if (k333 && d->CA == 8 && d->stride == 1) PLANE(8, 2, 3, 3, 3, 1);
        if (k333 && d->CA == 8 && d->stride == 2) PLANE(8, 2, 3, 3, 3, 2);
        if (k333 && d->CA == 16 && d->stride == 1) PLANE(16, 2, 3, 3, 3, 1);
        if (k333 && d->CA == 16 && d->stride == 2) PLANE(16, 2, 3, 3, 3, 2);
        // 8 position channels, stride 2: the plane-shift packing (DSH) fills the 8 idle matrix rows -- 2 instead of 3 / 4 tap tiles
        if (d->KD == 5 && d->KH == 3 && d->KW == 3 && d->CA == 8 && d->stride == 2 && d->CB == 8)
            { int r_ = launch_rows<8, 2, 5, 3, 3, 2, false, true>(d, a, b, in_scale, in_shift, ws, dw, s, ws_only, accumulate); if (r_ >= 0) return r_; }
        if (d->KD == 4 && d->KH == 4 && d->KW == 4 && d->CA == 8 && d->stride == 2 && d->CB == 8)
            { int r_ = launch_rows<8, 2, 4, 4, 4, 2, false, true>(d, a, b, in_scale, in_shift, ws, dw, s, ws_only, accumulate); if (r_ >= 0) return r_; }
        if (d->KD == 5 && d->KH == 3 && d->KW == 3 && d->CA == 8 && d->stride == 2) PLANE(8, 3, 5, 3, 3, 2);
        if (d->KD == 4 && d->KH == 4 && d->KW == 4 && d->CA == 8 && d->stride == 2) PLANE(8, 4, 4, 4, 4, 2);
    }
#undef PLANE
    if (d->CB == 8 && d->CA == 1 && k333 && d->stride == 1 && d->pad_d == 0 && d->pad_h == 0 && d->pad_w == 0 &&
        d->AD >= d->PD + 2 && d->AH >= d->PH + 2)
        return launch_wide(d, a, b, in_scale, in_shift, ws, dw, s, ws_only, accumulate);
    // (the first-generation LDS-tile MFMA kernel, wgrad_mfma_k, is kept above for reference but no longer instantiated:
    //  every geometry it served goes to wgrad_rows_k / wgrad_plane_k)
#define OWN(CB, CBT, KD, KH, KW, S, TPW, OKH) \
    return launch_own<CB, CBT, KD, KH, KW, S, TPW, OKH>(d, a, b, in_scale, in_shift, ws, dw, s, ws_only, accumulate)
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

static int grouped_dispatch(const vg_wgrad_desc* d, const float* a, const float* b, const float* in_scale, const float* in_shift,
                            float* ws, float* out, hipStream_t s, int64_t* ws_only) {
    if (!d) { vg_set_error("vg_wgrad3d_grouped: null descriptor"); return VG_ERR_ARG; }
    if (d->N <= 0 || d->per_group <= 0 || d->N % d->per_group || d->CB <= 0 || d->CB >= 16 || d->PD <= 0 || d->PH <= 0 || d->PW <= 0) {
        vg_set_error("vg_wgrad3d_grouped: bad shape"); return VG_ERR_ARG;
    }
    const bool k333 = d->KD == 3 && d->KH == 3 && d->KW == 3;
    if (k333 && d->CA == 1 && d->stride == 1 && !d->pad_d && !d->pad_h && !d->pad_w && d->PW <= 128) {
        int r_ = launch_rows<1, 2, 3, 3, 3, 1, false>(d, a, b, in_scale, in_shift, ws, out, s, ws_only, 0, 1);
        if (r_ >= 0) return r_;
    }
    vg_set_error("vg_wgrad3d_grouped: no kernel instance for CB=%d CA=%d k=%dx%dx%d stride=%d", d->CB, d->CA, d->KD, d->KH, d->KW, d->stride);
    return VG_ERR_UNSUPPORTED;
}

extern "C" int64_t vg_wgrad3d_grouped_ws_bytes(const vg_wgrad_desc* d) {
    int64_t bytes = 0;
    int rc = grouped_dispatch(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &bytes);
    return rc ? -1 : bytes;
}

extern "C" int vg_wgrad3d_grouped(const vg_wgrad_desc* d, const float* a, const float* b, const float* in_scale,
                                  const float* in_shift, float* ws, float* out, void* stream) {
    if (!a || !b || !ws || !out || !in_scale || !in_shift) { vg_set_error("vg_wgrad3d_grouped: null argument"); return VG_ERR_ARG; }
    return grouped_dispatch(d, a, b, in_scale, in_shift, ws, out, (hipStream_t)stream, nullptr);
}

extern "C" int64_t vg_wgrad3d_ws_bytes(const vg_wgrad_desc* d) {
    int64_t bytes = 0;
    int rc = dispatch(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &bytes, 0);
    return rc ? -1 : bytes;
}

extern "C" int vg_wgrad3d(const vg_wgrad_desc* d, const float* a, const float* b, const float* in_scale,
                          const float* in_shift, float* ws, float* dw, int32_t accumulate, void* stream) {
    if (!a || !b || !ws || !dw) { vg_set_error("vg_wgrad3d: null argument"); return VG_ERR_ARG; }
    return dispatch(d, a, b, in_scale, in_shift, ws, dw, (hipStream_t)stream, nullptr, accumulate);
}
