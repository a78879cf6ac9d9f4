// Latent sample + KL of the low-rank Gaussian posterior, and the ELBO assembly, as single launches.
//
// The reference does these with ~35 + ~45 elementwise/reduction operators per step (vae_reg_GP.py:321-329, 339-342,
// 400, 406-410).  On a 32 x 32 latent every one of them is a ~2 us kernel on the step's critical path, so the
// arithmetic is fused here: one launch forward, one backward, a few KB of traffic each.
//
//   d      = exp(a) + 1e-6 * [any(exp(a) < 1e-6)]                      (:321-323, batch-wide floor)
//   z      = mu + w * eps_w + sqrt(d) * eps_d                          (:325, LowRankMultivariateNormal.rsample)
//   kl     = 0.5 * ( -log(1 + sum w^2/d) - sum log d + sum d + sum w^2 + sum mu^2 - L )          (:400)
//   zcat   = [z, onehot(g)] for the G = C+1 decoder variants                                  (:326-329, 339-342)
#include "vg_common.h"

namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// wave per row, 4 rows per block; every block scans all of `a` for the batch-wide floor flag itself (B*L is ~1 K numbers:
// cheaper than a second launch or a grid-wide handshake)
__global__ void __launch_bounds__(256)
latent_fwd_k(const float* __restrict__ mu, const float* __restrict__ w, const float* __restrict__ a,
             const float* __restrict__ eps_w, const float* __restrict__ eps_d, int B, int L, int G,
             float* __restrict__ zcat, float* __restrict__ kl, float* __restrict__ d_out, float* __restrict__ flag_out) {
    __shared__ int s_flag;
    const int tid = threadIdx.x, lane = tid % VG_WAVE, wave = vg_wave_id(), nw = blockDim.x / VG_WAVE;
    if (tid == 0) s_flag = 0;
    __syncthreads();
    int f = 0;
    for (int i = tid; i < B * L; i += blockDim.x) f |= (expf(a[i]) < 1e-6f) ? 1 : 0;
    if (f) s_flag = 1;                                   // benign race: every writer stores 1
    __syncthreads();
    const float floor_ = s_flag ? 1e-6f : 0.f;
    if (tid == 0 && blockIdx.x == 0) flag_out[0] = floor_;
    const int Z = L + G;
    for (int b = blockIdx.x * nw + wave; b < B; b += gridDim.x * nw) {
        float s_wd = 0.f, s_ld = 0.f, s_d = 0.f, s_w = 0.f, s_m = 0.f;
        const float ew = eps_w[b];
        for (int l = lane; l < L; l += VG_WAVE) {
            const int i = b * L + l;
            const float d = expf(a[i]) + floor_;
            const float m = mu[i], ww = w[i];
            const float z = m + ww * ew + sqrtf(d) * eps_d[i];
            d_out[i] = d;
            for (int g = 0; g < G; ++g) zcat[((size_t)g * B + b) * Z + l] = z;
            s_wd += ww * ww / d; s_ld += logf(d); s_d += d; s_w += ww * ww; s_m += m * m;
        }
        s_wd = wsum(s_wd); s_ld = wsum(s_ld); s_d = wsum(s_d); s_w = wsum(s_w); s_m = wsum(s_m);
        if (lane == 0) kl[b] = 0.5f * (-(logf(1.f + s_wd) + s_ld) + s_d + s_w + s_m - (float)L);
        for (int j = lane; j < G * G; j += VG_WAVE) {
            const int g = j / G, c = j % G;
            zcat[((size_t)g * B + b) * Z + L + c] = (g == c) ? 1.f : 0.f;
        }
    }
}

__global__ void __launch_bounds__(256)
latent_bwd_k(const float* __restrict__ mu, const float* __restrict__ w, const float* __restrict__ d_in,
             const float* __restrict__ flag, const float* __restrict__ eps_w, const float* __restrict__ eps_d,
             const float* __restrict__ g_zcat, const float* __restrict__ g_kl, int B, int L, int G,
             float* __restrict__ g_mu, float* __restrict__ g_w, float* __restrict__ g_a) {
    const int lane = threadIdx.x % VG_WAVE, wave = vg_wave_id(), nw = blockDim.x / VG_WAVE;
    const int Z = L + G;
    const float floor_ = flag[0];
    for (int b = blockIdx.x * nw + wave; b < B; b += gridDim.x * nw) {
        float s_wd = 0.f;
        for (int l = lane; l < L; l += VG_WAVE) { const int i = b * L + l; s_wd += w[i] * w[i] / d_in[i]; }
        const float cap = 1.f + wsum(s_wd);
        const float gk = g_kl ? g_kl[b] : 0.f, ew = eps_w[b];
        for (int l = lane; l < L; l += VG_WAVE) {
            const int i = b * L + l;
            float gz = 0.f;
            if (g_zcat) for (int g = 0; g < G; ++g) gz += g_zcat[((size_t)g * B + b) * Z + l];
            const float d = d_in[i], m = mu[i], ww = w[i];
            g_mu[i] = gz + gk * m;
            g_w[i] = gz * ew + gk * (ww - ww / (d * cap));
            const float gd = gz * eps_d[i] * 0.5f / sqrtf(d) + gk * 0.5f * (ww * ww / (d * d * cap) - 1.f / d + 1.f);
            g_a[i] = gd * (d - floor_);                  // d(d)/d(a) = exp(a)
        }
    }
}

// loss = coef0 * sum kl + coef1 * sum slp + coef2 * gp_kl + coef3 * sum dist          (:406-410)
__global__ void __launch_bounds__(256)
loss_fwd_k(const float* __restrict__ kl, const float* __restrict__ slp, const float* __restrict__ dist,
           const float* __restrict__ gp_kl, int B, int CB, float c0, float c1, float c2, float c3, float* __restrict__ loss) {
    __shared__ float red[3][4];
    const int tid = threadIdx.x, lane = tid % VG_WAVE, wave = vg_wave_id();
    float a = 0.f, b = 0.f, c = 0.f;
    for (int i = tid; i < B; i += blockDim.x) { a += kl[i]; b += slp[i]; }
    for (int i = tid; i < CB; i += blockDim.x) c += dist[i];
    a = wsum(a); b = wsum(b); c = wsum(c);
    if (lane == 0) { red[0][wave] = a; red[1][wave] = b; red[2][wave] = c; }
    __syncthreads();
    if (tid == 0) {
        float sa = 0.f, sb = 0.f, sc = 0.f;
        for (int k = 0; k < (int)(blockDim.x / VG_WAVE); ++k) { sa += red[0][k]; sb += red[1][k]; sc += red[2][k]; }
        loss[0] = c0 * sa + c1 * sb + c2 * gp_kl[0] + c3 * sc;
    }
}

__global__ void __launch_bounds__(256)
loss_bwd_k(const float* __restrict__ g, int B, int CB, float c0, float c1, float c2, float c3,
           float* __restrict__ g_kl, float* __restrict__ g_slp, float* __restrict__ g_dist, float* __restrict__ g_gp) {
    const float go = g[0];
    for (int i = threadIdx.x; i < B; i += blockDim.x) { g_kl[i] = go * c0; g_slp[i] = go * c1; }
    for (int i = threadIdx.x; i < CB; i += blockDim.x) g_dist[i] = go * c3;
    if (threadIdx.x == 0) g_gp[0] = go * c2;
}

}  // namespace

extern "C" int vg_latent_fwd(const float* mu, const float* w, const float* a, const float* eps_w, const float* eps_d,
                             int32_t B, int32_t L, int32_t G, float* zcat, float* kl, float* d_out, float* flag_out,
                             void* stream) {
    if (!mu || !w || !a || !eps_w || !eps_d || !zcat || !kl || !d_out || !flag_out) { vg_set_error("vg_latent_fwd: null argument"); return VG_ERR_ARG; }
    if (B <= 0 || L <= 0 || G <= 0 || (int64_t)B * L > (1 << 24)) { vg_set_error("vg_latent_fwd: bad shape"); return VG_ERR_ARG; }
    vg_launch(latent_fwd_k, dim3((B + 3) / 4 < 64 ? (B + 3) / 4 : 64), dim3(256), 0, (hipStream_t)stream, mu, w, a, eps_w, eps_d, (int)B, (int)L, (int)G, zcat, kl,
              d_out, flag_out);
    return vg_check_launch("latent_fwd");
}

extern "C" int vg_latent_bwd(const float* mu, const float* w, const float* d, const float* flag, const float* eps_w,
                             const float* eps_d, const float* g_zcat, const float* g_kl, int32_t B, int32_t L, int32_t G,
                             float* g_mu, float* g_w, float* g_a, void* stream) {
    if (!mu || !w || !d || !flag || !eps_w || !eps_d || !g_mu || !g_w || !g_a) { vg_set_error("vg_latent_bwd: null argument"); return VG_ERR_ARG; }
    if (B <= 0 || L <= 0 || G <= 0 || (int64_t)B * L > (1 << 24)) { vg_set_error("vg_latent_bwd: bad shape"); return VG_ERR_ARG; }
    vg_launch(latent_bwd_k, dim3((B + 3) / 4 < 64 ? (B + 3) / 4 : 64), dim3(256), 0, (hipStream_t)stream, mu, w, d, flag, eps_w, eps_d, g_zcat, g_kl, (int)B, (int)L,
              (int)G, g_mu, g_w, g_a);
    return vg_check_launch("latent_bwd");
}

extern "C" int vg_loss_fwd(const float* kl, const float* slp, const float* dist, const float* gp_kl, int32_t B, int32_t CB,
                           double c_kl, double c_slp, double c_gp, double c_dist, float* loss, void* stream) {
    if (!kl || !slp || !dist || !gp_kl || !loss || B <= 0 || CB < 0) { vg_set_error("vg_loss_fwd: bad argument"); return VG_ERR_ARG; }
    vg_launch(loss_fwd_k, dim3(1), dim3(256), 0, (hipStream_t)stream, kl, slp, dist, gp_kl, (int)B, (int)CB, (float)c_kl,
              (float)c_slp, (float)c_gp, (float)c_dist, loss);
    return vg_check_launch("loss_fwd");
}

extern "C" int vg_loss_bwd(const float* g_loss, int32_t B, int32_t CB, double c_kl, double c_slp, double c_gp, double c_dist,
                           float* g_kl, float* g_slp, float* g_dist, float* g_gp, void* stream) {
    if (!g_loss || !g_kl || !g_slp || !g_dist || !g_gp || B <= 0 || CB < 0) { vg_set_error("vg_loss_bwd: bad argument"); return VG_ERR_ARG; }
    vg_launch(loss_bwd_k, dim3(1), dim3(256), 0, (hipStream_t)stream, g_loss, (int)B, (int)CB, (float)c_kl, (float)c_slp,
              (float)c_gp, (float)c_dist, g_kl, g_slp, g_dist, g_gp);
    return vg_check_launch("loss_bwd");
}
