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
__global__ void __launch_bounds__(64)
bn_fold_k(const double* __restrict__ part, int GC, int chunks, double count, int nout, double* __restrict__ sums) {
    const int i = blockIdx.x, lane = threadIdx.x;
    double a = 0, b = 0;
    for (int k = lane; k < chunks; k += VG_WAVE) { a += part[((size_t)i * chunks + k) * 2]; b += part[((size_t)i * chunks + k) * 2 + 1]; }
    a = wave_sum(a); b = wave_sum(b);
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

// fold + finalize in one launch (single-GPU path: nothing to all-reduce between them): one wavefront per (group, channel)
__global__ void __launch_bounds__(64)
bn_fold_finalize_k(const double* __restrict__ part, int C, int chunks, double count, const float* __restrict__ gamma,
                   const float* __restrict__ beta, float eps, float* __restrict__ scale, float* __restrict__ shift,
                   float* __restrict__ mean, float* __restrict__ rstd) {
    const int i = blockIdx.x, lane = threadIdx.x;
    double a = 0, b = 0;
    for (int k = lane; k < chunks; k += VG_WAVE) { a += part[((size_t)i * chunks + k) * 2]; b += part[((size_t)i * chunks + k) * 2 + 1]; }
    a = wave_sum(a); b = wave_sum(b);
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
__global__ void __launch_bounds__(64)
csum_fold_k(const double* __restrict__ part, int G, int C, int chunks, int accumulate, float* __restrict__ out) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double a = 0;
    for (int g = 0; g < G; ++g)
        for (int k = lane; k < chunks; k += VG_WAVE) a += part[((size_t)g * C + c) * chunks + k];
    a = wave_sum(a);
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
        vg_launch(bn_fold_finalize_k, dim3(G * C), dim3(64), 0, s, (const double*)part, (int)C, chunks, (double)total, gamma, beta, eps,
                  scale, shift, mean, rstd);
        return vg_check_launch("bn_fold_finalize");
    }
    vg_launch(bn_fold_k, dim3(G * C), dim3(64), 0, s, (const double*)part, G * C, chunks, (double)total, 3, sums);
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
    vg_launch(bn_fold_k, dim3(G * C), dim3(64), 0, s, (const double*)part, G * C, chunks, (double)total, 2, sums);
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
        vg_launch(csum_fold_k, dim3(C), dim3(64), 0, (hipStream_t)stream, (const double*)csum_part, G, (int)C, pl.chunks(),
                  (int)chsum_accumulate, chsum);
        return vg_check_launch("bn_bwd csum_fold");
    }
    return VG_OK;
}

namespace {
__global__ void __launch_bounds__(64)
chsum_fold_k(const double* __restrict__ part, int C, int chunks, int accumulate, float* __restrict__ out) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double a = 0;
    for (int k = lane; k < chunks; k += VG_WAVE) a += part[((size_t)c * chunks + k) * 2];
    a = wave_sum(a);
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
    vg_launch(chsum_fold_k, dim3(C), dim3(64), 0, s, (const double*)part, (int)C, chunks, (int)accumulate, out);
    return vg_check_launch("channel_sum fold");
}

extern "C" int vg_bn_stats_from_parts(const double* part, int32_t G, int32_t C, int64_t chunks, double count,
                                      const float* gamma, const float* beta, float eps, double* ext_sums, double* sums_ws,
                                      float* scale, float* shift, float* mean, float* rstd, void* stream) {
    if (!part || G <= 0 || C <= 0 || chunks <= 0 || !(count > 0)) { vg_set_error("vg_bn_stats_from_parts: bad argument"); return VG_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    if (!ext_sums) {
        if (!scale || !shift || !mean || !rstd) { vg_set_error("vg_bn_stats_from_parts: null output"); return VG_ERR_ARG; }
        vg_launch(bn_fold_finalize_k, dim3(G * C), dim3(64), 0, s, part, (int)C, (int)chunks, count, gamma, beta, eps, scale, shift, mean, rstd);
        return vg_check_launch("bn_fold_finalize(parts)");
    }
    vg_launch(bn_fold_k, dim3(G * C), dim3(64), 0, s, part, G * C, (int)chunks, count, 3, ext_sums);
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
}  // namespace

extern "C" int vg_bn_param_grad(const double* sums, int32_t G, int32_t C, float* dgamma, float* dbeta, int32_t accumulate,
                                void* stream) {
    if (!sums || !dgamma || !dbeta || G <= 0 || C <= 0) { vg_set_error("vg_bn_param_grad: bad argument"); return VG_ERR_ARG; }
    vg_launch(bn_param_grad_k, dim3(vg_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, sums, (int)G, (int)C, (int)accumulate, dgamma,
              dbeta);
    return vg_check_launch("bn_param_grad");
}
