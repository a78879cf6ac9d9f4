// vg_gam.hip -- fused GAM accumulate + Gaussian log-likelihood + GLM distance, and fused Adam (gfx950).
//
// Replaces, per minibatch (vae_reg_GP.py):
//   cons_i = task_var_i[:,None] * sigmoid(logits_i)   (:380, with the decoder's sigmoid :264 folded in)
//   x_rec  = sigmoid(logits_0) + sum_i cons_i         (:330, :390)
//   dist   = || cons_i[b] - glm_i ||_2                (:388; sum(cdist(..)) == B * sum_b dist)
//   slp[b] = sum_v log N(x[b,v] | x_rec[b,v], exp(-eps[v]))   (:401-405)
// in ONE pass over the (C+1) x B x V logits instead of 3-4 elementwise passes per covariate plus
// the (C+2) x B x V device-to-host copies, and the matching backward in one more pass.
// HBM-streaming kernels: coalesced loads, per-wave shuffle reductions, fixed-order second stage.
#include "vg_common.h"
#include "../../include/vaegam.h"

namespace {

constexpr int GT = 256;          // threads per block
constexpr int GV = 4;            // voxels per thread (strided by GT -> coalesced dword loads)
constexpr int GMAXC = 16;        // covariates supported by the LDS reduction scratch
constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;

__device__ __forceinline__ float wsum(float v) { return vg_wave_sum(v); }     // (every lane gets the sum; callers use lane 0's)
__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }

// grid (vchunks, B).  part_slp[b][chunk], part_d2[i][b][chunk]
__global__ void __launch_bounds__(GT)
gam_fwd_k(const float* __restrict__ logits, const float* __restrict__ gain, const float* __restrict__ x,
          const double* __restrict__ eps, const float* __restrict__ glm, int C, int B, long long V,
          float* __restrict__ part_slp, float* __restrict__ part_d2, float* __restrict__ maps_out) {
    __shared__ float red[GT / VG_WAVE][GMAXC + 1];
    const int chunk = blockIdx.x, b = blockIdx.y, chunks = gridDim.x;
    const int lane = threadIdx.x % VG_WAVE, wave = threadIdx.x / VG_WAVE;
    const long long v0 = (long long)chunk * GT * GV + threadIdx.x;
    float xrec[GV];
    bool ok[GV];
#pragma unroll
    for (int k = 0; k < GV; ++k) { ok[k] = (v0 + (long long)k * GT) < V; xrec[k] = 0.f; }
    const size_t BV = (size_t)B * V;
    // base map
#pragma unroll
    for (int k = 0; k < GV; ++k)
        if (ok[k]) {
            const long long v = v0 + (long long)k * GT;
            const float s = sigmoidf_(logits[(size_t)b * V + v]);
            xrec[k] = s;
            if (maps_out) maps_out[(size_t)b * V + v] = s;
        }
    for (int i = 1; i <= C; ++i) {
        const float gn = gain[(size_t)(i - 1) * B + b];
        float d2 = 0.f;
#pragma unroll
        for (int k = 0; k < GV; ++k)
            if (ok[k]) {
                const long long v = v0 + (long long)k * GT;
                const float cons = gn * sigmoidf_(logits[(size_t)i * BV + (size_t)b * V + v]);
                const float df = cons - glm[(size_t)(i - 1) * V + v];
                d2 = fmaf(df, df, d2);
                xrec[k] += cons;
                if (maps_out) maps_out[(size_t)i * BV + (size_t)b * V + v] = cons;
            }
        d2 = wsum(d2);
        if (lane == 0) red[wave][i] = d2;
    }
    float lp = 0.f;
#pragma unroll
    for (int k = 0; k < GV; ++k)
        if (ok[k]) {
            const long long v = v0 + (long long)k * GT;
            const float sg = (float)exp(-eps[v]);               // fp64 exp then cast, as :402
            const float r = x[(size_t)b * V + v] - xrec[k];
            lp += -(r * r) / (2.f * sg * sg) - logf(sg) - LOG_SQRT_2PI;   // Normal.log_prob
            if (maps_out) maps_out[(size_t)(C + 1) * BV + (size_t)b * V + v] = xrec[k];
        }
    lp = wsum(lp);
    if (lane == 0) red[wave][0] = lp;
    __syncthreads();
    if (threadIdx.x <= C) {
        float t = 0.f;
        for (int w = 0; w < GT / VG_WAVE; ++w) t += red[w][threadIdx.x];
        if (threadIdx.x == 0) part_slp[(size_t)b * chunks + chunk] = t;
        else part_d2[((size_t)(threadIdx.x - 1) * B + b) * chunks + chunk] = t;
    }
}

// one thread per output: slp[b] = sum chunks; dist[i][b] = sqrt(sum chunks)
__global__ void __launch_bounds__(64)
gam_fwd_fold_k(const float* __restrict__ part_slp, const float* __restrict__ part_d2, int C, int B,
               int chunks, float* __restrict__ slp, float* __restrict__ dist) {
    const int i = blockIdx.x, lane = threadIdx.x;                          // one wavefront per output
    if (i >= (C + 1) * B) return;
    const float* src = (i < B) ? part_slp + (size_t)i * chunks : part_d2 + (size_t)(i - B) * chunks;
    double a = 0;
    for (int k = lane; k < chunks; k += VG_WAVE) a += src[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
    if (lane == 0) { if (i < B) slp[i] = (float)a; else dist[i - B] = (float)sqrt(a); }
}

// backward.  grid (vblocks, BS): block handles voxels [vb*GT, +GT) and samples b = bs, bs+BS, ...
//   part_dsig[bs][v], part_dgain[i][b][vb*4 + wave]
// Every logit is read ONCE and its sigmoid kept in a register for the second pass (CMAX = compile-time bound of the covariate
// loop, fully unrolled); the per-(covariate, sample) sums of d gain leave the block as one partial per WAVE, so the sample loop has
// no barrier at all (the first version re-read and re-evaluated all C+1 sigmoids and paid two __syncthreads per covariate and
// sample: 228 us at batch 64 / 8 covariates against 60 us of HBM time).
template <int CMAX>
__global__ void __launch_bounds__(GT)
gam_bwd_k(const float* __restrict__ logits, const float* __restrict__ gain, const float* __restrict__ x,
          const double* __restrict__ eps, const float* __restrict__ glm, const float* __restrict__ dist,
          const float* __restrict__ g_slp, const float* __restrict__ g_dist, int C, int B, long long V,
          float* __restrict__ d_logits, float* __restrict__ part_dsig, float* __restrict__ part_dgain, float* __restrict__ part_tot) {
    const int vb = blockIdx.x, bs = blockIdx.y, BS = gridDim.y, nparts = gridDim.x * (GT / VG_WAVE);
    const int lane = threadIdx.x % VG_WAVE, wave = threadIdx.x / VG_WAVE;
    const long long v = (long long)vb * GT + threadIdx.x;
    const bool ok = v < V;
    const long long vc = ok ? v : V - 1;                               // clamped: loads stay unconditional, results are masked
    const size_t BV = (size_t)B * V;
    const float sg = (float)exp(-eps[vc]), inv_var = 1.f / (sg * sg);
    float glm_v[CMAX];
#pragma unroll
    for (int i = 0; i < CMAX; ++i) glm_v[i] = i < C ? glm[(size_t)i * V + vc] : 0.f;
    float dsig = 0.f, tot = 0.f;                                       // tot: sum of every d_logits element this thread writes
    for (int b = bs; b < B; b += BS) {
        const size_t row = (size_t)b * V + vc;
        float sgm[CMAX + 1], gn[CMAX];
        sgm[0] = logits[row];
#pragma unroll
        for (int i = 0; i < CMAX; ++i) {
            sgm[i + 1] = i < C ? logits[(size_t)(i + 1) * BV + row] : 0.f;
            gn[i] = i < C ? gain[(size_t)i * B + b] : 0.f;
        }
        const float xv = x[row], gs = g_slp[b];
#pragma unroll
        for (int i = 0; i <= CMAX; ++i) sgm[i] = sigmoidf_(sgm[i]);
        float xr = sgm[0];
#pragma unroll
        for (int i = 0; i < CMAX; ++i) xr += gn[i] * sgm[i + 1];         // (gn = 0 beyond C)
        const float r = xv - xr;
        const float dxr = gs * r * inv_var;                            // d slp / d x_rec = r / sigma^2
        if (ok) {
            dsig += gs * (r * r * inv_var / sg - 1.f / sg);            // d slp / d sigma
            const float dl = dxr * sgm[0] * (1.f - sgm[0]);
            d_logits[row] = dl; tot += dl;
        }
#pragma unroll
        for (int i = 0; i < CMAX; ++i) {
            if (i < C) {                                               // wave-uniform
                const float s_ = sgm[i + 1];
                const float dd = dist[(size_t)i * B + b];
                float dcons = dxr;
                if (dd > 0.f) dcons += g_dist[(size_t)i * B + b] * (gn[i] * s_ - glm_v[i]) / dd;
                const float dl = dcons * gn[i] * s_ * (1.f - s_);
                if (ok) { d_logits[(size_t)(i + 1) * BV + row] = dl; tot += dl; }
                const float dg = wsum(ok ? dcons * s_ : 0.f);
                if (lane == 0) part_dgain[((size_t)i * B + b) * nparts + vb * (GT / VG_WAVE) + wave] = dg;
            }
        }
    }
    if (ok) part_dsig[(size_t)bs * V + v] = dsig;
    if (part_tot) {
        tot = wsum(tot);
        if (lane == 0) part_tot[((size_t)bs * gridDim.x + vb) * (GT / VG_WAVE) + wave] = tot;
    }
}

// d_total (+)= sum of the per-wave totals (one block)
__global__ void __launch_bounds__(256)
gam_bwd_fold_total_k(const float* __restrict__ part_tot, int n, int accumulate, float* __restrict__ d_total) {
    __shared__ double red[256 / VG_WAVE];
    double a = 0;
    for (int k = threadIdx.x; k < n; k += 256) a += part_tot[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
    if (threadIdx.x % VG_WAVE == 0) red[threadIdx.x / VG_WAVE] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int w = 0; w < 256 / VG_WAVE; ++w) t += red[w];
        d_total[0] = (accumulate ? d_total[0] : 0.f) + (float)t;
    }
}

// one wavefront per (covariate, sample) entry: lanes stride over the per-block partials, then a shuffle reduction
// (one THREAD per entry walked ~V/256 dependent loads: 65 us for 96 numbers)
__global__ void __launch_bounds__(64)
gam_bwd_fold_gain_k(const float* __restrict__ part_dgain, int CB, int vblocks, float* __restrict__ d_gain) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= CB) return;
    double a = 0;
    for (int k = lane; k < vblocks; k += VG_WAVE) a += part_dgain[(size_t)i * vblocks + k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
    if (lane == 0) d_gain[i] = (float)a;
}

// d_eps[v] = dL/dsigma (fp32 sum over samples, cast to fp64) * dsigma/deps = -exp(-eps) (fp64)
__global__ void gam_bwd_fold_eps_k(const float* __restrict__ part_dsig, const double* __restrict__ eps, int BS, long long V,
                                   double* __restrict__ d_eps) {
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    float a = 0.f;
    for (int k = 0; k < BS; ++k) a += part_dsig[(size_t)k * V + v];
    d_eps[v] = (double)a * (-exp(-eps[v]));
}

int fwd_chunks(long long V) { return (int)((V + (long long)GT * GV - 1) / ((long long)GT * GV)); }
int bwd_vblocks(long long V) { return (int)((V + GT - 1) / GT); }
int bwd_gain_parts(long long V) { return bwd_vblocks(V) * (GT / VG_WAVE); }
int bwd_bs(int B) { return B < 8 ? B : 8; }

// ---------------------------------------------------------------------------------- Adam
template <typename T>
__global__ void __launch_bounds__(256)
adam_k(T* __restrict__ p, const T* __restrict__ g, T* __restrict__ m, T* __restrict__ v, long long n,
       double b1, double b2, double eps, const double* __restrict__ sc) {
    const T step_size = (T)sc[0], bc2_sqrt = (T)sc[1];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const T gi = g[i];
        const T mi = m[i] + (gi - m[i]) * (T)(1.0 - b1);               // lerp_(grad, 1-beta1)
        const T vi = v[i] * (T)b2 + (T)(1.0 - b2) * gi * gi;           // mul_(beta2).addcmul_(g, g, 1-beta2)
        m[i] = mi; v[i] = vi;
        const T denom = sqrt(vi) / bc2_sqrt + (T)eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

// One thread: advance the optimiser's step count ON THE DEVICE and derive the two bias-correction scalars adam_k reads.
// st = [step_size, bc2_sqrt, t].  Living inside the (captured) step, the count can never run ahead of the launches that
// use it -- a host-staged scalar buffer rewritten for step t+1 could be read by step t's still-queued kernels.
__global__ void adam_advance_k(double* __restrict__ st, double lr, double b1, double b2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double t = st[2] + 1.0;
    st[2] = t;
    st[0] = lr / (1.0 - pow(b1, t));
    st[1] = sqrt(1.0 - pow(b2, t));
}

// ---------------------------------------------------------------------------------- weight packing
// all conv / transposed-conv weights of the model -> the [ci][tap][co] images the conv kernels read through the scalar
// path (forward and data-gradient variants), in ONE launch straight from the flat parameter buffer
__global__ void __launch_bounds__(256)
pack_weights_k(const float* __restrict__ flat, float* __restrict__ packed, const long long* __restrict__ gsegs, int nseg, long long total) {
    // the segment table (a few hundred bytes) is searched per element: from LDS, not as a chain of dependent global loads
    constexpr int MAXSEG = 64;
    __shared__ long long ssegs[MAXSEG * 8];
    const bool in_lds = nseg <= MAXSEG;
    if (in_lds) for (int i = threadIdx.x; i < nseg * 8; i += blockDim.x) ssegs[i] = gsegs[i];
    __syncthreads();
    const long long* segs = in_lds ? ssegs : gsegs;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        int sgi = 0;
        while (sgi + 1 < nseg && e >= segs[(sgi + 1) * 8 + 1]) ++sgi;            // segments are sorted by destination offset
        const long long* sg = segs + sgi * 8;
        const long long src = sg[0], dst = sg[1];
        const int d0 = (int)sg[2], d1 = (int)sg[3], kvol = (int)sg[4], mode = (int)sg[5];
        const long long r = e - dst;
        if (r >= sg[6]) continue;            // alignment gap behind the segment
        long long si;
        if (mode == 0) {                     // out[i1][t][i0] = w[i0][i1][t]
            const int i0 = (int)(r % d0); const long long q = r / d0; const int t = (int)(q % kvol); const int i1 = (int)(q / kvol);
            si = ((long long)i0 * d1 + i1) * kvol + t;
        } else {                             // out[i0][t][i1] = w[i0][i1][t or kvol-1-t]
            const int i1 = (int)(r % d1); const long long q = r / d1; const int t = (int)(q % kvol); const int i0 = (int)(q / kvol);
            si = ((long long)i0 * d1 + i1) * kvol + (mode == 2 ? kvol - 1 - t : t);
        }
        packed[e] = flat[src + si];
    }
}

}  // namespace

extern "C" int vg_pack_weights(const float* flat, float* packed, const int64_t* segs, int32_t nseg, int64_t total, void* stream) {
    if (!flat || !packed || !segs || nseg <= 0 || total <= 0) { vg_set_error("vg_pack_weights: bad arguments"); return VG_ERR_ARG; }
    long long blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    vg_launch(pack_weights_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, flat, packed, (const long long*)segs, (int)nseg, (long long)total);
    return vg_check_launch("pack_weights");
}

extern "C" int64_t vg_gam_ws_bytes(int32_t C, int32_t B, int64_t V) {
    if (C < 0 || C > GMAXC || B <= 0 || V <= 0) return -1;
    const int64_t fwd = ((int64_t)B * fwd_chunks(V) + (int64_t)C * B * fwd_chunks(V)) * sizeof(float);
    const int64_t bwd = ((int64_t)bwd_bs(B) * V + (int64_t)C * B * bwd_gain_parts(V) + (int64_t)bwd_bs(B) * bwd_gain_parts(V)) * sizeof(float);
    return fwd > bwd ? fwd : bwd;
}

extern "C" int vg_gam_elbo_fwd(const float* logits, const float* gain, const float* x, const double* eps,
                               const float* glm, int32_t C, int32_t B, int64_t V, void* ws,
                               float* sum_log_prob, float* dist, float* maps_out, void* stream) {
    if (!logits || !x || !eps || !ws || !sum_log_prob || (C > 0 && (!gain || !glm || !dist)) || C < 0 || C > GMAXC || B <= 0 ||
        B > 65535 || V <= 0) {
        vg_set_error("vg_gam_elbo_fwd: bad arguments C=%d B=%d V=%lld", C, B, (long long)V); return VG_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const int chunks = fwd_chunks(V);
    float* part_slp = (float*)ws;
    float* part_d2 = part_slp + (size_t)B * chunks;
    vg_launch(gam_fwd_k, dim3(chunks, B), dim3(GT), 0, s, logits, gain, x, eps, glm, (int)C, (int)B, (long long)V, part_slp, part_d2, maps_out);
    int rc = vg_check_launch("gam_fwd");
    if (rc) return rc;
    vg_launch(gam_fwd_fold_k, dim3((C + 1) * B), dim3(64), 0, s, (const float*)part_slp, (const float*)part_d2,
              (int)C, (int)B, chunks, sum_log_prob, dist);
    return vg_check_launch("gam_fwd_fold");
}

extern "C" int vg_gam_elbo_bwd(const float* logits, const float* gain, const float* x, const double* eps,
                               const float* glm, const float* dist, const float* g_slp, const float* g_dist,
                               int32_t C, int32_t B, int64_t V, void* ws,
                               float* d_logits, float* d_gain, double* d_eps, float* d_total, int32_t total_accumulate, void* stream) {
    if (!logits || !x || !eps || !ws || !g_slp || !d_logits || !d_eps || (C > 0 && (!gain || !glm || !dist || !g_dist || !d_gain)) ||
        C < 0 || C > GMAXC || B <= 0 || V <= 0) {
        vg_set_error("vg_gam_elbo_bwd: bad arguments C=%d B=%d V=%lld", C, B, (long long)V); return VG_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const int vblocks = bwd_vblocks(V), BS = bwd_bs(B);
    float* part_dsig = (float*)ws;
    float* part_dgain = part_dsig + (size_t)BS * V;
    float* part_tot = d_total ? part_dgain + (size_t)C * B * bwd_gain_parts(V) : nullptr;
    if (C <= 8)
        vg_launch(gam_bwd_k<8>, dim3(vblocks, BS), dim3(GT), 0, s, logits, gain, x, eps, glm, dist, g_slp, g_dist, (int)C, (int)B,
                  (long long)V, d_logits, part_dsig, part_dgain, part_tot);
    else if (C <= 16)
        vg_launch(gam_bwd_k<16>, dim3(vblocks, BS), dim3(GT), 0, s, logits, gain, x, eps, glm, dist, g_slp, g_dist, (int)C, (int)B,
                  (long long)V, d_logits, part_dsig, part_dgain, part_tot);
    else { vg_set_error("vg_gam_elbo_bwd: more than 16 covariates"); return VG_ERR_UNSUPPORTED; }
    int rc = vg_check_launch("gam_bwd");
    if (rc) return rc;
    if (C > 0) {
        vg_launch(gam_bwd_fold_gain_k, dim3(C * B), dim3(64), 0, s, (const float*)part_dgain, (int)(C * B), bwd_gain_parts(V), d_gain);
        if ((rc = vg_check_launch("gam_bwd_fold_gain"))) return rc;
    }
    vg_launch(gam_bwd_fold_eps_k, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, s, (const float*)part_dsig, eps, BS, (long long)V, d_eps);
    if ((rc = vg_check_launch("gam_bwd_fold_eps"))) return rc;
    if (d_total) {
        vg_launch(gam_bwd_fold_total_k, dim3(1), dim3(256), 0, s, (const float*)part_tot, BS * bwd_gain_parts(V), (int)total_accumulate, d_total);
        return vg_check_launch("gam_bwd_fold_total");
    }
    return VG_OK;
}

extern "C" int vg_adam_advance(double* state, double lr, double b1, double b2, void* stream) {
    if (!state || !(lr > 0) || !(b1 >= 0 && b1 < 1) || !(b2 >= 0 && b2 < 1)) { vg_set_error("vg_adam_advance: bad arguments"); return VG_ERR_ARG; }
    vg_launch(adam_advance_k, dim3(1), dim3(64), 0, (hipStream_t)stream, state, lr, b1, b2);
    return vg_check_launch("adam_advance");
}

extern "C" int vg_adam_step(void* p, const void* g, void* m, void* v, int64_t n, int32_t is_f64,
                            double b1, double b2, double eps, const double* step_scalars, void* stream) {
    if (!p || !g || !m || !v || !step_scalars || n <= 0) { vg_set_error("vg_adam_step: bad arguments"); return VG_ERR_ARG; }
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipStream_t s = (hipStream_t)stream;
    if (is_f64)
        vg_launch(adam_k<double>, dim3((unsigned)blocks), dim3(256), 0, s, (double*)p, (const double*)g, (double*)m, (double*)v,
                  (long long)n, b1, b2, eps, step_scalars);
    else
        vg_launch(adam_k<float>, dim3((unsigned)blocks), dim3(256), 0, s, (float*)p, (const float*)g, (float*)m, (float*)v,
                  (long long)n, b1, b2, eps, step_scalars);
    return vg_check_launch("adam");
}
