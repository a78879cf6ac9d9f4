// vg_bn.hip -- batch-norm batch statistics, their backward, and per-channel sums (gfx950).
//
// BatchNorm3d(track_running_stats=False) (vae_reg_GP.py:194-196, 216-218) normalises with the
// statistics of the current minibatch in train AND eval mode.  The decoder is launched for all
// C+1 one-hot variants at once, so statistics are kept per "group" (sample n -> group
// n / per_group).  Pure HBM streaming: block partials are reduced through wavefront shuffles,
// combined in double by a second small kernel (fixed order, no float atomics), and the raw
// [sum, sumsq, count] triples can be handed to the caller for a cross-rank all-reduce.
#include "vg_common.h"
#include "../../include/vaegam.h"

namespace {

constexpr int BN_THREADS = 256;
constexpr int BN_EPT = 16;                       // elements per thread per block pass

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// block-level sum of two doubles; result valid in thread 0
__device__ __forceinline__ void block_sum2(double& a, double& b) {
    __shared__ double red[2][BN_THREADS / VG_WAVE];
    const int lane = threadIdx.x % VG_WAVE, wave = threadIdx.x / VG_WAVE;
    a = wave_sum(a); b = wave_sum(b);
    __syncthreads();                              // red[] may still be read from a previous call
    if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = 0; b = 0;
        for (int w = 0; w < BN_THREADS / VG_WAVE; ++w) { a += red[0][w]; b += red[1][w]; }
    }
}

// grid (chunks, C, G).  part[((g*C + c)*chunks + chunk)*2 + {0,1}]
// MODE 0: sum h, sum h^2 (h = relu?(x));  MODE 1: sum dxe, sum dxe*hhat  (hhat = (relu?(p)-mean)*rstd)
template <int MODE>
__global__ void __launch_bounds__(BN_THREADS)
bn_partial_k(const float* __restrict__ x, const float* __restrict__ p, const float* __restrict__ mean,
             const float* __restrict__ rstd, int C, long long P, int per_group, int relu, int cp, double* __restrict__ part) {
    const int chunk = blockIdx.x, c = blockIdx.y, g = blockIdx.z, chunks = gridDim.x;
    const int cpi = chunk % cp, si = chunk / cp, ns = chunks / cp;      // position chunk, sample split
    float mu = 0.f, rs = 1.f;
    if (MODE == 1) { mu = mean[g * C + c]; rs = rstd[g * C + c]; }
    float s0 = 0.f, s1 = 0.f;
    double d0 = 0.0, d1 = 0.0;
    int since = 0;
    for (int nn = si; nn < per_group; nn += ns) {
        const long long base = (((long long)g * per_group + nn) * C + c) * P;
        for (long long e = (long long)cpi * BN_THREADS + threadIdx.x; e < P; e += (long long)cp * BN_THREADS) {
            const long long off = base + e;
            if (MODE == 0) {
                float v = x[off];
                if (relu) v = fmaxf(v, 0.f);
                s0 += v; s1 = fmaf(v, v, s1);
            } else {
                float h = p[off];
                if (relu) h = fmaxf(h, 0.f);
                const float dv = x[off];
                s0 += dv; s1 = fmaf(dv, (h - mu) * rs, s1);
            }
            if (++since == 64) { d0 += s0; d1 += s1; s0 = 0.f; s1 = 0.f; since = 0; }   // bound fp32 run length
        }
    }
    d0 += s0; d1 += s1;
    block_sum2(d0, d1);
    if (threadIdx.x == 0) {
        part[((size_t)(g * C + c) * chunks + chunk) * 2 + 0] = d0;
        part[((size_t)(g * C + c) * chunks + chunk) * 2 + 1] = d1;
    }
}

// one WAVE per (g,c): fold the chunk partials in a fixed order (lane-strided partial sums, then the shuffle tree);
// nout = 3 writes [s0, s1, count], nout = 2 writes [s0, s1]
__global__ void __launch_bounds__(BN_THREADS)
bn_fold_k(const double* __restrict__ part, int GC, int chunks, double count, int nout, double* __restrict__ sums) {
    const int i = blockIdx.x, lane = threadIdx.x;
    double a = 0, b = 0;
    for (int k = lane; k < chunks; k += BN_THREADS) { a += part[((size_t)i * chunks + k) * 2]; b += part[((size_t)i * chunks + k) * 2 + 1]; }
    block_sum2(a, b);
    if (lane == 0) {
        sums[(size_t)i * nout] = a; sums[(size_t)i * nout + 1] = b;
        if (nout == 3) sums[(size_t)i * nout + 2] = count;
    }
}

__global__ void bn_finalize_k(const double* __restrict__ sums, int G, int C, const float* __restrict__ gamma,
                              const float* __restrict__ beta, float eps, float* __restrict__ scale,
                              float* __restrict__ shift, float* __restrict__ mean, float* __restrict__ rstd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G * C) return;
    const int c = i % C;
    const double cnt = sums[(size_t)i * 3 + 2];
    const double mu = sums[(size_t)i * 3] / cnt;
    double var = sums[(size_t)i * 3 + 1] / cnt - mu * mu;    // biased variance, as F.batch_norm
    if (var < 0) var = 0;
    const float rs = (float)(1.0 / sqrt(var + (double)eps));
    const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float sc = gm * rs;
    scale[i] = sc; shift[i] = bt - (float)mu * sc; mean[i] = (float)mu; rstd[i] = rs;
}

// fold + finalize in one launch (single-GPU path: nothing to all-reduce between them): one block per (group, channel)
// (256 threads: with one wavefront the ~1300 partials of a large layer were 21 dependent loads deep -- 16 us for 72 numbers)
__global__ void __launch_bounds__(BN_THREADS)
bn_fold_finalize_k(const double* __restrict__ part, int C, int chunks, double count, const float* __restrict__ gamma,
                   const float* __restrict__ beta, float eps, float* __restrict__ scale, float* __restrict__ shift,
                   float* __restrict__ mean, float* __restrict__ rstd) {
    const int i = blockIdx.x, lane = threadIdx.x;
    double a = 0, b = 0;
    for (int k = lane; k < chunks; k += BN_THREADS) { a += part[((size_t)i * chunks + k) * 2]; b += part[((size_t)i * chunks + k) * 2 + 1]; }
    block_sum2(a, b);
    if (lane == 0) {
        const int c = i % C;
        const double mu = a / count;
        double var = b / count - mu * mu;                       // biased variance, as F.batch_norm
        if (var < 0) var = 0;
        const float rs = (float)(1.0 / sqrt(var + (double)eps));
        const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        const float sc = gm * rs;
        scale[i] = sc; shift[i] = bt - (float)mu * sc; mean[i] = (float)mu; rstd[i] = rs;
    }
}

// grid (chunks, C, G): dp = relu'(p) * gamma*rstd * (dxe - m1 - hhat*m2), in place over dxe
__global__ void __launch_bounds__(BN_THREADS)
bn_bwd_apply_k(float* __restrict__ dxe, const float* __restrict__ p, int C, long long P, int per_group, int relu,
               const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
               const double* __restrict__ sums, double count, int cp, float* __restrict__ dgamma_part, float* __restrict__ dbeta_part,
               double* __restrict__ csum_part) {
    const int chunk = blockIdx.x, c = blockIdx.y, g = blockIdx.z, chunks = gridDim.x;
    const int cpi = chunk % cp, si = chunk / cp, ns = chunks / cp;
    const int gc = g * C + c;
    const float mu = mean[gc], rs = rstd[gc];
    const float m1 = (float)(sums[(size_t)gc * 2] / count), m2 = (float)(sums[(size_t)gc * 2 + 1] / count);
    const float k = (gamma ? gamma[c] : 1.f) * rs;
    if (chunk == 0 && threadIdx.x == 0) {
        dbeta_part[gc] = (float)sums[(size_t)gc * 2];
        dgamma_part[gc] = (float)sums[(size_t)gc * 2 + 1];
    }
    float vs = 0.f;
    for (int nn = si; nn < per_group; nn += ns) {
        const long long base = (((long long)g * per_group + nn) * C + c) * P;
        for (long long e = (long long)cpi * BN_THREADS + threadIdx.x; e < P; e += (long long)cp * BN_THREADS) {
            const long long off = base + e;
            const float pv = p[off];
            const float h = relu ? fmaxf(pv, 0.f) : pv;
            const float hh = (h - mu) * rs;
            float v = k * (dxe[off] - m1 - hh * m2);
            if (relu && !(pv > 0.f)) v = 0.f;
            dxe[off] = v;
            vs += v;
        }
    }
    if (csum_part) {                              // per-channel sum of the result = the producing layer's bias gradient
        double a = (double)vs, b = 0.0;
        block_sum2(a, b);
        if (threadIdx.x == 0) csum_part[(size_t)gc * chunks + chunk] = a;
    }
}

// out[c] (+)= sum over groups and chunks of part[(g*C + c)*chunks + k]
__global__ void __launch_bounds__(BN_THREADS)
csum_fold_k(const double* __restrict__ part, int G, int C, int chunks, int accumulate, float* __restrict__ out) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double a = 0, unused = 0;
    for (int g = 0; g < G; ++g)
        for (int k = lane; k < chunks; k += BN_THREADS) a += part[((size_t)g * C + c) * chunks + k];
    block_sum2(a, unused);
    if (lane == 0) out[c] = (accumulate ? out[c] : 0.f) + (float)a;
}

// chunks over the P positions of one sample (cp) and sample splits (ns): grid.x = cp * ns, sized so
// that the launch has >= ~2048 blocks when the tensor is large enough to want them
struct BnPlan { int cp, ns; int chunks() const { return cp * ns; } };
BnPlan plan_for(long long P, int per_group, int C, int G) {
    BnPlan b;
    long long c = (P + (long long)BN_THREADS * BN_EPT - 1) / ((long long)BN_THREADS * BN_EPT);
    if (c < 1) c = 1;
    if (c > 64) c = 64;
    b.cp = (int)c;
    long long want = 2048 / ((long long)b.cp * C * G);
    if (want < 1) want = 1;
    if (want > per_group) want = per_group;
    if (want > 64) want = 64;
    b.ns = (int)want;
    return b;
}

}  // namespace

extern "C" int64_t vg_bn_ws_bytes(int32_t N, int32_t C, int64_t P, int32_t per_group) {
    if (N <= 0 || C <= 0 || P <= 0 || per_group <= 0 || N % per_group) return -1;
    const int G = N / per_group;
    const int chunks = plan_for(P, per_group, C, G).chunks();
    return (int64_t)G * C * chunks * 2 * sizeof(double) + (int64_t)G * C * 3 * sizeof(double);
}

static int bn_args_ok(const char* who, const void* x, int N, int C, long long P, int per_group) {
    if (!x || N <= 0 || C <= 0 || P <= 0 || per_group <= 0 || N % per_group) {
        vg_set_error("%s: bad arguments N=%d C=%d P=%lld per_group=%d", who, N, C, P, per_group); return VG_ERR_ARG;
    }
    if (C > 65535 || N / per_group > 65535) { vg_set_error("%s: grid limit", who); return VG_ERR_ARG; }
    return VG_OK;
}

extern "C" int vg_bn_finalize(const double* sums, int32_t G, int32_t C, const float* gamma, const float* beta,
                              float eps, float* scale, float* shift, float* mean, float* rstd, void* stream) {
    if (!sums || !scale || !shift || !mean || !rstd || G <= 0 || C <= 0) { vg_set_error("vg_bn_finalize: bad arguments"); return VG_ERR_ARG; }
    vg_launch(bn_finalize_k, dim3(vg_cdiv(G * C, 64)), dim3(64), 0, (hipStream_t)stream, sums, G, C, gamma, beta, eps, scale, shift, mean, rstd);
    return vg_check_launch("bn_finalize");
}

extern "C" int vg_bn_stats(const float* x, int32_t N, int32_t C, int64_t P, int32_t per_group, int32_t relu,
                           const float* gamma, const float* beta, float eps, void* ws, double* ext_sums,
                           float* scale, float* shift, float* mean, float* rstd, void* stream) {
    int rc = bn_args_ok("vg_bn_stats", x, N, C, P, per_group);
    if (rc) return rc;
    if (!ws) { vg_set_error("vg_bn_stats: null workspace"); return VG_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const int G = N / per_group;
    const long long total = (long long)per_group * P;
    const BnPlan pl = plan_for(P, per_group, C, G);
    const int chunks = pl.chunks();
    double* part = (double*)ws;
    double* sums = ext_sums ? ext_sums : part + (size_t)G * C * chunks * 2;
    vg_launch(bn_partial_k<0>, dim3(chunks, C, G), dim3(BN_THREADS), 0, s, x, (const float*)nullptr, (const float*)nullptr,
              (const float*)nullptr, (int)C, (long long)P, (int)per_group, (int)relu, pl.cp, part);
    if ((rc = vg_check_launch("bn_partial"))) return rc;
    if (!ext_sums) {
        if (!scale || !shift || !mean || !rstd) { vg_set_error("vg_bn_stats: null output"); return VG_ERR_ARG; }
        vg_launch(bn_fold_finalize_k, dim3(G * C), dim3(BN_THREADS), 0, s, (const double*)part, (int)C, chunks, (double)total, gamma, beta, eps,
                  scale, shift, mean, rstd);
        return vg_check_launch("bn_fold_finalize");
    }
    vg_launch(bn_fold_k, dim3(G * C), dim3(BN_THREADS), 0, s, (const double*)part, G * C, chunks, (double)total, 3, sums);
    return vg_check_launch("bn_fold");                // caller all-reduces, then calls vg_bn_finalize
}

extern "C" int vg_bn_bwd_reduce(const float* dxe, const float* p, int32_t N, int32_t C, int64_t P, int32_t per_group,
                                int32_t relu, const float* mean, const float* rstd, void* ws, double* sums, void* stream) {
    int rc = bn_args_ok("vg_bn_bwd_reduce", dxe, N, C, P, per_group);
    if (rc) return rc;
    if (!p || !mean || !rstd || !ws || !sums) { vg_set_error("vg_bn_bwd_reduce: null argument"); return VG_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const int G = N / per_group;
    const long long total = (long long)per_group * P;
    const BnPlan pl = plan_for(P, per_group, C, G);
    const int chunks = pl.chunks();
    double* part = (double*)ws;
    vg_launch(bn_partial_k<1>, dim3(chunks, C, G), dim3(BN_THREADS), 0, s, dxe, p, mean, rstd, (int)C, (long long)P,
              (int)per_group, (int)relu, pl.cp, part);
    if ((rc = vg_check_launch("bn_bwd_partial"))) return rc;
    vg_launch(bn_fold_k, dim3(G * C), dim3(BN_THREADS), 0, s, (const double*)part, G * C, chunks, (double)total, 2, sums);
    return vg_check_launch("bn_bwd_fold");
}

extern "C" int vg_bn_bwd_apply(float* dxe, const float* p, int32_t N, int32_t C, int64_t P, int32_t per_group,
                               int32_t relu, const float* gamma, const float* mean, const float* rstd,
                               const double* sums, double count, float* dgamma_part, float* dbeta_part,
                               void* ws, float* chsum, int32_t chsum_accumulate, void* stream) {
    int rc = bn_args_ok("vg_bn_bwd_apply", dxe, N, C, P, per_group);
    if (rc) return rc;
    if (!p || !mean || !rstd || !sums || !dgamma_part || !dbeta_part || !(count > 0)) { vg_set_error("vg_bn_bwd_apply: bad argument"); return VG_ERR_ARG; }
    if (chsum && !ws) { vg_set_error("vg_bn_bwd_apply: chsum needs the vg_bn_ws_bytes workspace"); return VG_ERR_ARG; }
    const int G = N / per_group;
    const BnPlan pl = plan_for(P, per_group, C, G);
    double* csum_part = chsum ? (double*)ws : nullptr;
    vg_launch(bn_bwd_apply_k, dim3(pl.chunks(), C, G), dim3(BN_THREADS), 0, (hipStream_t)stream, dxe, p, (int)C, (long long)P,
              (int)per_group, (int)relu, gamma, mean, rstd, sums, count, pl.cp, dgamma_part, dbeta_part, csum_part);
    if ((rc = vg_check_launch("bn_bwd_apply"))) return rc;
    if (chsum) {
        vg_launch(csum_fold_k, dim3(C), dim3(BN_THREADS), 0, (hipStream_t)stream, (const double*)csum_part, G, (int)C, pl.chunks(),
                  (int)chsum_accumulate, chsum);
        return vg_check_launch("bn_bwd csum_fold");
    }
    return VG_OK;
}

namespace {
// ------------------------------------------------------------------------------------------------------------------
// Batch-norm backward FUSED with the data gradient of the layer behind it, for the decoder's last stage
// (bnt5 -> convt5, vae_reg_GP.py:218,264: ConvTranspose3d(C, 1, 3, stride 1) on the largest activation of the network).
// The gradient reaching the batch norm,  dxe[n][c][q] = sum_k dy[n][q + k] * w[c][k]  (27 taps of a ONE-channel tensor),
// is cheaper to recompute from dy than to store and re-read: both passes below form it on the fly from a dy tile staged in
// LDS, so the C-channel gradient tensor is never written (one full-size write and two full-size reads less per step),
// and the separate data-gradient launch disappears.
//   pass 1 (reduce): reads dy + p           -> per (group, channel) [sum dxe, sum dxe*hhat]
//   pass 2 (apply) : reads dy + p, writes dp = relu'(p) * gamma*rstd * (dxe - m1 - hhat*m2)   (+ per-channel sum of dp)
// grid (d-tiles, N): a block owns TD planes of one sample's p (all rows / columns); it stages the TD+2 dy planes they touch --
// ONE contiguous span -- by flat LDS-DMA; a thread walks flattened positions (coalesced loads of p, stores of dp).
constexpr int FT_TD = 3;                         // planes of p per block (39 = 13 x 3; 80 = 26 x 3 + 2)
constexpr int FT_MAXC = 16;

struct FtParams {
    int N, C, ID, IH, IW;                        // p: [N][C][ID][IH][IW];  dy: [N][1][ID+2][IH+2][IW+2]
    int per_group, relu, tilesD;
    float inv_iw, inv_plane;                     // 1/IW, 1/(IH*IW) for the position decode
};

// a wave-uniform value pinned to a VECTOR register: the per-channel constants below are uniform, the compiler would keep all of them in
// scalar registers next to the 27 weights of the running channel, run out of SGPRs and spill them into VGPR lanes (v_readlane per use)
#ifdef VG_EMU
static inline float to_vgpr(float v) { return v; }
#else
__device__ __forceinline__ float to_vgpr(float v) { float r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(v)); return r; }
#endif

// MODE 0: reduce, MODE 1: apply
template <int MODE, int CT>
__global__ void __launch_bounds__(BN_THREADS)
bn_tconv1_k(const float* __restrict__ dy, const float* __restrict__ w, const float* __restrict__ p, float* __restrict__ dp,
            const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
            const double* __restrict__ sums, double count, FtParams a, double* __restrict__ part, double* __restrict__ csum_part) {
    VG_DYN_SMEM(float, tile);
    const int tid = threadIdx.x, lane = tid % VG_WAVE, wave = vg_wave_id();
    const int n = blockIdx.y, td = blockIdx.x;
    const int C = CT ? CT : a.C;
    const int g = n / a.per_group;
    const int OH = a.IH + 2, OW = a.IW + 2, oplane = OH * OW, iplane = a.IH * a.IW;
    const int d0 = td * FT_TD, nd = min(FT_TD, a.ID - d0);
    // ---- stage dy planes [d0, d0 + nd + 2): contiguous
    {
        const float* src = dy + ((size_t)n * (a.ID + 2) + d0) * oplane;
        const int nfl = (nd + 2) * oplane;
        vg_dma_block(src, tile, nfl, wave, BN_THREADS / VG_WAVE, lane);
        vg_dma_wait();
    }
    __syncthreads();
    float mu[CT ? CT : FT_MAXC], rs[CT ? CT : FT_MAXC], k1[CT ? CT : FT_MAXC], m1[CT ? CT : FT_MAXC], m2[CT ? CT : FT_MAXC];
    float acc0[CT ? CT : FT_MAXC], acc1[CT ? CT : FT_MAXC];
#pragma unroll
    for (int c = 0; c < (CT ? CT : FT_MAXC); ++c) {
        if (c < C) {
            const int gc = g * C + c;
            mu[c] = to_vgpr(mean[gc]); rs[c] = to_vgpr(rstd[gc]);
            if (MODE == 1) {
                m1[c] = to_vgpr((float)(sums[(size_t)gc * 2] / count)); m2[c] = to_vgpr((float)(sums[(size_t)gc * 2 + 1] / count));
                k1[c] = to_vgpr((gamma ? gamma[c] : 1.f) * rstd[gc]);
            }
        }
        acc0[c] = 0.f; acc1[c] = 0.f;
    }
    const int npos = nd * iplane;
    const float lo = a.relu ? 0.f : -__builtin_inff();
    const size_t vol = (size_t)a.ID * iplane;
    const float* pb = p + (size_t)n * C * vol + (size_t)d0 * iplane;
    float* dpb = MODE == 1 ? dp + (size_t)n * C * vol + (size_t)d0 * iplane : nullptr;
    // PP positions per thread and iteration (strided by the block: every load / store stays coalesced): the 27 weights of a channel
    // are fetched through the scalar path once per iteration and used for PP positions.  All C*27 weights do not fit the scalar
    // registers; left to itself the compiler hoists them out of the loop anyway and spills them into VGPR lanes (210 v_readlane
    // per position in the first version of this kernel) -- the pointer is laundered per channel so that they are re-loaded.
    constexpr int PP = 2;
    for (int e0 = tid; e0 < npos; e0 += PP * BN_THREADS) {
        int ee[PP]; bool ok[PP]; int toff[PP];
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            const int e = e0 + u * BN_THREADS;
            ok[u] = e < npos;
            ee[u] = ok[u] ? e : e0;
            // e -> (dl, h, x): float reciprocal estimate + one correction step (exact for these sizes)
            int dl = (int)((float)ee[u] * a.inv_plane);
            int r = ee[u] - dl * iplane;
            dl += (r >= iplane ? 1 : 0) - (r < 0 ? 1 : 0);                  // branch-free correction of the estimate
            r = ee[u] - dl * iplane;
            int h = (int)((float)r * a.inv_iw);
            int x = r - h * a.IW;
            h += (x >= a.IW ? 1 : 0) - (x < 0 ? 1 : 0);
            x = r - h * a.IW;
            toff[u] = dl * oplane + h * OW + x;
        }
        float pv[PP][CT ? CT : FT_MAXC];
#pragma unroll
        for (int u = 0; u < PP; ++u)
#pragma unroll
            for (int c = 0; c < (CT ? CT : FT_MAXC); ++c) if (c < C) pv[u][c] = pb[(size_t)c * vol + ee[u]];
        // the 27 dy values each position's gradient is made of (shared by all channels)
        float win[PP][27];
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            const float* t0 = tile + toff[u];
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const float* row = t0 + kd * oplane + kh * OW;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) win[u][(kd * 3 + kh) * 3 + kw] = row[kw];
                }
        }
        // channel loop, software-pipelined by hand: the 27 weights of channel c+1 are requested (scalar loads) before the FMAs of
        // channel c; sched_barrier keeps the compiler from pulling ALL channels' loads to the top of the iteration (216 scalar
        // registers do not exist: it then spills them through VGPR lanes)
        float wcur[27], wnxt[27];
        {
            vg_cptr w0 = VG_CPTR(w);
#pragma unroll
            for (int k = 0; k < 27; ++k) wcur[k] = w0[k];
        }
#pragma unroll
        for (int c = 0; c < (CT ? CT : FT_MAXC); ++c) {
            if (c >= C) continue;
#ifndef VG_EMU
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (c + 1 < C) {
                int coff = (c + 1) * 27;
#ifndef VG_EMU
                asm volatile("" : "+s"(coff));
#endif
                vg_cptr wn = VG_CPTR(w) + coff;
#pragma unroll
                for (int k = 0; k < 27; ++k) wnxt[k] = wn[k];
            }
            float dxe[PP];
#pragma unroll
            for (int u = 0; u < PP; ++u) dxe[u] = 0.f;
#pragma unroll
            for (int k = 0; k < 27; ++k) {
#pragma unroll
                for (int u = 0; u < PP; ++u) dxe[u] = fmaf(win[u][k], wcur[k], dxe[u]);
            }
#pragma unroll
            for (int u = 0; u < PP; ++u) {
                const float hv = vg_max(pv[u][c], lo);
                const float hh = (hv - mu[c]) * rs[c];
                if (MODE == 0) {
                    const float dm = ok[u] ? dxe[u] : 0.f;
                    acc0[c] += dm; acc1[c] = fmaf(dm, hh, acc1[c]);
                } else {
                    float v = k1[c] * (dxe[u] - m1[c] - hh * m2[c]);
                    v = (pv[u][c] > lo) ? v : 0.f;                     // ReLU backward (lo = -inf without a ReLU: always true)
                    if (ok[u]) { dpb[(size_t)c * vol + ee[u]] = v; acc0[c] += v; }
                }
            }
#pragma unroll
            for (int k = 0; k < 27; ++k) wcur[k] = wnxt[k];
        }
    }
    // ---- block partials (fixed order): MODE 0 -> part[((g*C+c)*chunks + chunk)*2 + {0,1}];  MODE 1 -> csum_part[(g*C+c)*chunks + chunk]
    __shared__ double red[2][FT_MAXC][BN_THREADS / VG_WAVE];
    const int chunks = a.per_group * a.tilesD;
    const int chunk = (n % a.per_group) * a.tilesD + td;
#pragma unroll
    for (int c = 0; c < (CT ? CT : FT_MAXC); ++c) {
        if (c >= C) continue;
        const double s0 = wave_sum((double)acc0[c]);
        const double s1 = MODE == 0 ? wave_sum((double)acc1[c]) : 0.0;
        if (lane == 0) { red[0][c][wave] = s0; red[1][c][wave] = s1; }
    }
    __syncthreads();
    if (tid < C) {
        double s0 = 0, s1 = 0;
        for (int wv = 0; wv < BN_THREADS / VG_WAVE; ++wv) { s0 += red[0][tid][wv]; s1 += red[1][tid][wv]; }
        const size_t gc = (size_t)g * C + tid;
        if (MODE == 0) { part[(gc * chunks + chunk) * 2] = s0; part[(gc * chunks + chunk) * 2 + 1] = s1; }
        else if (csum_part) csum_part[gc * chunks + chunk] = s0;
    }
}

static int ft_setup(const char* who, const float* dy, const float* w, const float* p, int N, int C, int ID, int IH, int IW, int per_group,
                    FtParams* a, size_t* shmem) {
    if (!dy || !w || !p || N <= 0 || C <= 0 || C > FT_MAXC || ID <= 0 || IH <= 0 || IW <= 0 || per_group <= 0 || N % per_group || N > 65535) {
        vg_set_error("%s: bad arguments N=%d C=%d p=%dx%dx%d per_group=%d", who, N, C, ID, IH, IW, per_group); return VG_ERR_ARG;
    }
    a->N = N; a->C = C; a->ID = ID; a->IH = IH; a->IW = IW; a->per_group = per_group; a->relu = 0;
    a->tilesD = vg_cdiv(ID, FT_TD);
    a->inv_iw = 1.0f / (float)IW; a->inv_plane = 1.0f / (float)(IH * IW);
    *shmem = (size_t)(FT_TD + 2) * (IH + 2) * (IW + 2) * sizeof(float) + 256;
    if (*shmem > 150 * 1024) { vg_set_error("%s: a %dx%d dy plane tile does not fit LDS", who, IH + 2, IW + 2); return VG_ERR_UNSUPPORTED; }
    if ((long long)FT_TD * IH * IW >= (1 << 22)) { vg_set_error("%s: tile too large for the position decode", who); return VG_ERR_UNSUPPORTED; }
    return VG_OK;
}

}  // namespace

extern "C" int64_t vg_bn_tconv1_ws_bytes(int32_t N, int32_t C, int32_t ID, int32_t per_group) {
    if (N <= 0 || C <= 0 || ID <= 0 || per_group <= 0 || N % per_group) return -1;
    const int64_t chunks = (int64_t)per_group * vg_cdiv(ID, FT_TD);
    return (int64_t)(N / per_group) * C * chunks * 2 * (int64_t)sizeof(double);
}

extern "C" int vg_bn_bwd_reduce_tconv1(const float* dy, const float* w, const float* p, int32_t N, int32_t C, int32_t ID, int32_t IH,
                                       int32_t IW, int32_t per_group, int32_t relu, const float* mean, const float* rstd, void* ws,
                                       double* sums, void* stream) {
    FtParams a; size_t shmem;
    int rc = ft_setup("vg_bn_bwd_reduce_tconv1", dy, w, p, N, C, ID, IH, IW, per_group, &a, &shmem);
    if (rc) return rc;
    if (!mean || !rstd || !ws || !sums) { vg_set_error("vg_bn_bwd_reduce_tconv1: null argument"); return VG_ERR_ARG; }
    a.relu = relu;
    hipStream_t s = (hipStream_t)stream;
    const int G = N / per_group, chunks = per_group * a.tilesD;
    double* part = (double*)ws;
    if (C == 8)
        vg_launch(bn_tconv1_k<0, 8>, dim3(a.tilesD, N), dim3(BN_THREADS), shmem, s, dy, w, p, (float*)nullptr, (const float*)nullptr, mean, rstd,
                  (const double*)nullptr, 1.0, a, part, (double*)nullptr);
    else
        vg_launch(bn_tconv1_k<0, 0>, dim3(a.tilesD, N), dim3(BN_THREADS), shmem, s, dy, w, p, (float*)nullptr, (const float*)nullptr, mean, rstd,
                  (const double*)nullptr, 1.0, a, part, (double*)nullptr);
    if ((rc = vg_check_launch("bn_bwd_reduce_tconv1"))) return rc;
    vg_launch(bn_fold_k, dim3(G * C), dim3(BN_THREADS), 0, s, (const double*)part, G * C, chunks, (double)per_group * ID * IH * IW, 2, sums);
    return vg_check_launch("bn_bwd_reduce_tconv1 fold");
}

// Batch-norm backward sums of the layer in front of a one-output-channel transposed conv WITHOUT a pass over the data.
// With dxe[c][q] = sum_k w[c][k] dy[q+k] (the conv's data gradient) and Q[g][c][k] = sum_{n in g, q} hhat[n][c][q] dy[n][q+k]
// (the conv's weight gradient taken against the NORMALISED input), S[g][k] = sum_{n in g, q} dy[n][q+k]:
//   sum dxe      = sum_k w[c][k] S[g][k]          sum dxe*hhat = sum_k w[c][k] Q[g][c][k]
//   dw[c][k]     = gamma[c] sum_g Q[g][c][k] + beta[c] sum_g S[g][k]      (the input of the conv is gamma*hhat + beta)
// q = vg_wgrad3d_grouped's output [G][C+1][K] (row C holds S).  One block; G*C and C*K are a few hundred.
__global__ void __launch_bounds__(256)
bn_tconv1_sums_k(const float* __restrict__ q, const float* __restrict__ w, const float* __restrict__ gamma,
                 const float* __restrict__ beta, int G, int C, int K, double* __restrict__ sums, float* __restrict__ dw, int accumulate) {
    for (int i = threadIdx.x; i < G * C; i += blockDim.x) {
        const int g = i / C, c = i % C;
        const float* Q = q + ((size_t)g * (C + 1) + c) * K;
        const float* S = q + ((size_t)g * (C + 1) + C) * K;
        double s0 = 0, s1 = 0;
        for (int k = 0; k < K; ++k) { const double wv = w[c * K + k]; s0 += wv * S[k]; s1 += wv * Q[k]; }
        sums[(size_t)i * 2] = s0; sums[(size_t)i * 2 + 1] = s1;
    }
    for (int j = threadIdx.x; j < C * K; j += blockDim.x) {
        const int c = j / K, k = j % K;
        double a = 0, b = 0;
        for (int g = 0; g < G; ++g) { a += q[((size_t)g * (C + 1) + c) * K + k]; b += q[((size_t)g * (C + 1) + C) * K + k]; }
        const float v = (float)((double)gamma[c] * a + (double)beta[c] * b);
        dw[j] = accumulate ? dw[j] + v : v;
    }
}

extern "C" int vg_bn_tconv1_sums(const float* q, const float* w, const float* gamma, const float* beta, int32_t G, int32_t C, int32_t K,
                                 double* sums, float* dw, int32_t accumulate, void* stream) {
    if (!q || !w || !gamma || !beta || !sums || !dw || G <= 0 || C <= 0 || K <= 0) { vg_set_error("vg_bn_tconv1_sums: bad argument"); return VG_ERR_ARG; }
    vg_launch(bn_tconv1_sums_k, dim3(1), dim3(256), 0, (hipStream_t)stream, q, w, gamma, beta, (int)G, (int)C, (int)K, sums, dw, (int)accumulate);
    return vg_check_launch("bn_tconv1_sums");
}

extern "C" int vg_bn_bwd_apply_tconv1(const float* dy, const float* w, const float* p, float* dp, int32_t N, int32_t C, int32_t ID,
                                      int32_t IH, int32_t IW, int32_t per_group, int32_t relu, const float* gamma, const float* mean,
                                      const float* rstd, const double* sums, double count, void* ws, float* chsum,
                                      int32_t chsum_accumulate, void* stream) {
    FtParams a; size_t shmem;
    int rc = ft_setup("vg_bn_bwd_apply_tconv1", dy, w, p, N, C, ID, IH, IW, per_group, &a, &shmem);
    if (rc) return rc;
    if (!dp || !mean || !rstd || !sums || !(count > 0) || (chsum && !ws)) { vg_set_error("vg_bn_bwd_apply_tconv1: bad argument"); return VG_ERR_ARG; }
    a.relu = relu;
    hipStream_t s = (hipStream_t)stream;
    const int G = N / per_group, chunks = per_group * a.tilesD;
    double* csum_part = chsum ? (double*)ws : nullptr;
    if (C == 8)
        vg_launch(bn_tconv1_k<1, 8>, dim3(a.tilesD, N), dim3(BN_THREADS), shmem, s, dy, w, p, dp, gamma, mean, rstd, sums, count, a,
                  (double*)nullptr, csum_part);
    else
        vg_launch(bn_tconv1_k<1, 0>, dim3(a.tilesD, N), dim3(BN_THREADS), shmem, s, dy, w, p, dp, gamma, mean, rstd, sums, count, a,
                  (double*)nullptr, csum_part);
    if ((rc = vg_check_launch("bn_bwd_apply_tconv1"))) return rc;
    if (chsum) {
        vg_launch(csum_fold_k, dim3(C), dim3(BN_THREADS), 0, s, (const double*)csum_part, G, (int)C, chunks, (int)chsum_accumulate, chsum);
        return vg_check_launch("bn_bwd_apply_tconv1 csum_fold");
    }
    return VG_OK;
}

namespace {
__global__ void __launch_bounds__(BN_THREADS)
chsum_fold_k(const double* __restrict__ part, int C, int chunks, int accumulate, float* __restrict__ out) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double a = 0, unused = 0;
    for (int k = lane; k < chunks; k += BN_THREADS) a += part[((size_t)c * chunks + k) * 2];
    block_sum2(a, unused);
    if (lane == 0) out[c] = (accumulate ? out[c] : 0.f) + (float)a;
}
}  // namespace

// x viewed as one group of N samples: per-channel sum (bias gradients)
extern "C" int vg_channel_sum(const float* x, int32_t N, int32_t C, int64_t P, void* ws, float* out, int32_t accumulate, void* stream) {
    int rc = bn_args_ok("vg_channel_sum", x, N, C, P, N);
    if (rc) return rc;
    if (!ws || !out) { vg_set_error("vg_channel_sum: null argument"); return VG_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const BnPlan pl = plan_for(P, N, C, 1);
    const int chunks = pl.chunks();
    double* part = (double*)ws;
    vg_launch(bn_partial_k<0>, dim3(chunks, C, 1), dim3(BN_THREADS), 0, s, x, (const float*)nullptr, (const float*)nullptr,
              (const float*)nullptr, (int)C, (long long)P, (int)N, 0, pl.cp, part);
    if ((rc = vg_check_launch("channel_sum partial"))) return rc;
    vg_launch(chsum_fold_k, dim3(C), dim3(BN_THREADS), 0, s, (const double*)part, (int)C, chunks, (int)accumulate, out);
    return vg_check_launch("channel_sum fold");
}

extern "C" int vg_bn_stats_from_parts(const double* part, int32_t G, int32_t C, int64_t chunks, double count,
                                      const float* gamma, const float* beta, float eps, double* ext_sums, double* sums_ws,
                                      float* scale, float* shift, float* mean, float* rstd, void* stream) {
    if (!part || G <= 0 || C <= 0 || chunks <= 0 || !(count > 0)) { vg_set_error("vg_bn_stats_from_parts: bad argument"); return VG_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    if (!ext_sums) {
        if (!scale || !shift || !mean || !rstd) { vg_set_error("vg_bn_stats_from_parts: null output"); return VG_ERR_ARG; }
        vg_launch(bn_fold_finalize_k, dim3(G * C), dim3(BN_THREADS), 0, s, part, (int)C, (int)chunks, count, gamma, beta, eps, scale, shift, mean, rstd);
        return vg_check_launch("bn_fold_finalize(parts)");
    }
    vg_launch(bn_fold_k, dim3(G * C), dim3(BN_THREADS), 0, s, part, G * C, (int)chunks, count, 3, ext_sums);
    return vg_check_launch("bn_fold(parts)");
}

namespace {
// dgamma[c] (+)= sum_g sums[g][c][1], dbeta[c] (+)= sum_g sums[g][c][0]
__global__ void __launch_bounds__(64)
bn_param_grad_k(const double* __restrict__ sums, int G, int C, int accumulate, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a = 0, b = 0;
    for (int g = 0; g < G; ++g) { b += sums[((size_t)g * C + c) * 2]; a += sums[((size_t)g * C + c) * 2 + 1]; }
    dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)a;
    dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)b;
}
// The first layer's batch norm sits on the DATA (vae_reg_GP.py:187-189, 236-238: bn1 on the input volume, one channel): its backward
// needs no gradient w.r.t. the input, and the normalisation is folded into conv1's weight gradient -- dw_hat is taken against
// xhat = (x - mean) * rstd, then  dw = gamma * dw_hat + beta * db,  dgamma = <w, dw_hat>,  dbeta = <sum_taps w, db>.  As torch
// expressions + autograd's accumulation that tail was 14 launches (~60 us) at the very end of the step with nothing beside it.
__global__ void __launch_bounds__(256)
data_bn_grads_k(const float* __restrict__ dw_hat, const float* __restrict__ db, const float* __restrict__ w,
                const float* __restrict__ gamma, const float* __restrict__ beta, int CO, int CI, int T, int acc,
                float* __restrict__ dw, float* __restrict__ dbias, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[2][256];
    const int tid = threadIdx.x, n = CO * CI * T;
    for (int i = tid; i < n; i += 256) {
        const int co = i / (CI * T), ci = (i / T) % CI;
        const float v = gamma[ci] * dw_hat[i] + beta[ci] * db[co];
        dw[i] = acc ? dw[i] + v : v;
    }
    for (int co = tid; co < CO; co += 256) dbias[co] = acc ? dbias[co] + db[co] : db[co];
    for (int ci = 0; ci < CI; ++ci) {
        float sg = 0.f, sb = 0.f;
        for (int i = tid; i < CO * T; i += 256) {
            const int co = i / T, t = i - co * T;
            const float wv = w[((size_t)co * CI + ci) * T + t];
            sg += wv * dw_hat[((size_t)co * CI + ci) * T + t];
            sb += wv * db[co];
        }
        red[0][tid] = sg; red[1][tid] = sb;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
            __syncthreads();
        }
        if (tid == 0) {
            dgamma[ci] = acc ? dgamma[ci] + red[0][0] : red[0][0];
            dbeta[ci] = acc ? dbeta[ci] + red[1][0] : red[1][0];
        }
        __syncthreads();
    }
}

__global__ void data_bn_nshift_k(const float* __restrict__ mean, const float* __restrict__ rstd, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = -(mean[i] * rstd[i]);
}

}  // namespace

extern "C" int vg_data_bn_nshift(const float* mean, const float* rstd, int32_t n, float* nshift, void* stream) {
    if (!mean || !rstd || !nshift || n <= 0) { vg_set_error("vg_data_bn_nshift: bad argument"); return VG_ERR_ARG; }
    vg_launch(data_bn_nshift_k, dim3(vg_cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, mean, rstd, (int)n, nshift);
    return vg_check_launch("data_bn_nshift");
}

extern "C" int vg_data_bn_grads(const float* dw_hat, const float* db, const float* w, const float* gamma, const float* beta,
                                int32_t CO, int32_t CI, int32_t taps, float* dw, float* dbias, float* dgamma, float* dbeta,
                                int32_t accumulate, void* stream) {
    if (!dw_hat || !db || !w || !gamma || !beta || !dw || !dbias || !dgamma || !dbeta || CO <= 0 || CI <= 0 || taps <= 0) {
        vg_set_error("vg_data_bn_grads: bad argument"); return VG_ERR_ARG;
    }
    vg_launch(data_bn_grads_k, dim3(1), dim3(256), 0, (hipStream_t)stream, dw_hat, db, w, gamma, beta, (int)CO, (int)CI, (int)taps,
              (int)accumulate, dw, dbias, dgamma, dbeta);
    return vg_check_launch("data_bn_grads");
}

extern "C" int vg_bn_param_grad(const double* sums, int32_t G, int32_t C, float* dgamma, float* dbeta, int32_t accumulate,
                                void* stream) {
    if (!sums || !dgamma || !dbeta || G <= 0 || C <= 0) { vg_set_error("vg_bn_param_grad: bad argument"); return VG_ERR_ARG; }
    vg_launch(bn_param_grad_k, dim3(vg_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, sums, (int)G, (int)C, (int)accumulate, dgamma,
              dbeta);
    return vg_check_launch("bn_param_grad");
}
