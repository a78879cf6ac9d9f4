/* vaegam.h -- C ABI of libvaegam_hip.so: the MI355X (gfx950) kernels under the VAE-GAM train step.
 *
 * The reference (dannyfa/VAE-GAM) is pure Python on PyTorch and has no FFI: every entry point
 * below replaces a group of ATen calls made from `VAE.forward` / `loss.backward()` /
 * `optimizer.step()`; the reference lines are cited per function.  Conventions:
 *   - every pointer is a DEVICE pointer into memory owned by the caller (PyTorch tensors);
 *     the library allocates nothing persistent and frees nothing;
 *   - every launch is asynchronous on the `stream` handed in (a hipStream_t passed as void*);
 *   - return value: 0 = OK, non-zero = vg_status; text via vg_last_error(); no exceptions;
 *   - activations are fp32, NCDHW, contiguous.  Layers store PRE-activation values; the
 *     consumer applies ReLU and the batch-norm affine when it loads them (`vg_prologue`).
 *   - "group": the decoder runs C+1 one-hot variants in one launch (sample n -> group
 *     n / per_group); batch-norm statistics are kept per group (vae_reg_GP.py:330,343 call
 *     decode() separately per variant, each with its own batch statistics).
 */
#ifndef VAEGAM_H
#define VAEGAM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vg_conv_desc {
    int32_t N;                 /* samples in this launch */
    int32_t CI, CO;            /* channels of the tensor read / written by THIS launch */
    int32_t ID, IH, IW;        /* spatial size of the tensor read */
    int32_t OD, OH, OW;        /* spatial size of the tensor written */
    int32_t KD, KH, KW;        /* kernel */
    int32_t stride;            /* 1 or 2 */
    int32_t pad_d, pad_h, pad_w; /* corr: leading zero padding of the input; tconv: ConvTranspose3d padding */
    int32_t relu_in;           /* prologue: max(x,0) on load */
    int32_t per_group;         /* samples per affine group for in_scale/in_shift ([N/per_group][CI]) */
} vg_conv_desc;

/* library identity / errors */
int         vg_version(void);
const char* vg_last_error(void);

/* y[n][co][o] = bias[co] + sum_{ci,k} P(x)[n][ci][o*stride + k - pad] * wpk[ci][k][co]
 * (zero outside the input).  P = prologue (ReLU, then x*in_scale[g][ci]+in_shift[g][ci]).
 * Replaces F.conv3d forward (vae_reg_GP.py:238-242), ConvTranspose3d stride-1 forward with a
 * flipped repacked kernel (:260,262,264) and the data gradients of ConvTranspose3d (autograd
 * of :260-264).  If mask_src != NULL the result is multiplied by (mask_src[same index] > 0):
 * the ReLU backward of the producer layer fused into the data-gradient epilogue. */
int vg_corr3d(const vg_conv_desc* d, const float* x, const float* wpk, const float* bias,
              const float* in_scale, const float* in_shift, const float* mask_src, float* y, void* stream);

/* stride-2 transposed convolution, gather form:
 * y[n][co][o] = bias[co] + sum_{ci,k : (o+pad-k) even} P(x)[n][ci][(o+pad-k)/2] * wpk[ci][k][co].
 * Replaces ConvTranspose3d stride-2 forward (vae_reg_GP.py:261,263) and the data gradient of
 * the stride-2 Conv3d layers (autograd of :239,241). mask_src as above. */
int vg_tconv3d_s2(const vg_conv_desc* d, const float* x, const float* wpk, const float* bias,
                  const float* in_scale, const float* in_shift, const float* mask_src, float* y, void* stream);
/* vg_tconv3d_s2 that ALSO accumulates the batch statistics the NEXT layer's BatchNorm3d needs (vae_reg_GP.py:216-218,
 * 254-264: bnt3 after convt2, bnt5 after convt4) while the outputs are still in registers: per (group, channel) partial
 * [sum, sum of squares] of relu?(y), one pair per wavefront, laid out for vg_bn_stats_from_parts:
 *   stats_part[((g*CO + c)*chunks + k)*2 + {0,1}],  chunks = vg_tconv3d_s2_stats_chunks(d, stats_per_group).
 * Saves the separate full-tensor read of vg_bn_stats. */
int64_t vg_tconv3d_s2_stats_chunks(const vg_conv_desc* d, int32_t stats_per_group);
int vg_tconv3d_s2_stats(const vg_conv_desc* d, const float* x, const float* wpk, const float* bias,
                        const float* in_scale, const float* in_shift, float* y, int32_t stats_per_group,
                        int32_t stats_relu, double* stats_part, void* stream);

/* Table-driven implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32): the forward passes and
 * data gradients of the conv / transposed-conv layers with 8 or 16 output channels (vae_reg_GP.py:189-215; replaces F.conv3d /
 * F.conv_transpose3d and their autograd data gradients, like vg_corr3d / vg_tconv3d_s2, for the launches that are real contractions).
 * The host describes the layer as a position grid + window-offset tables (vae_gam_amd/ops.py: mm_plan):
 *   16 matrix rows = (16 / CO) replicas x CO output channels;  columns = 16 consecutive positions (pd, ph, pw) of a block's grid of
 *   PD x PH x PW positions;  class q (1 for correlations, the (rd, rh) output parities of a stride-2 transposed conv) contributes
 *   ks[q] k-steps of 4 window offsets per input channel.
 *   a_img [sum_q CI*ks[q]][64]: A operands, lane l = row (l & 15), offset 4*step + (l >> 4)  -- gathered from the weights by vg_gather_f32;
 *   tau   [sum_q ks[q]][4] int32: the window offset dd*IH*IW + dh*IW + dw of (step, kk);  dlt [sum_q ks[q]][4][3] int32: (dd, dh, dw).
 *   position (pd, ph, pw) reads input element (pd*sdi + dd, ph*shi + dh, pw*swi + dw) (zero outside the tensor) and writes output
 *   (pd*sdo + od0[q], ph*sho + oh0[q], pw*swo + ow0[q] + replica).
 *   Block (bd, bh) of a sample works on position planes [bd*PD, +PD) x rows [bh*PHB, +PHB) and stages, cc channels at a time, the rows
 *   [bh*PHB*shi + hlo, ...+ (PHB-1)*shi + hhi] of input planes [bd*PD*sdi + d0, + LD) (LDS-DMA, one contiguous span per plane; dbuf: double-buffered).
 * bias / in_scale / in_shift / mask_src / stats_*: as vg_corr3d / vg_tconv3d_s2_stats (stats chunks: vg_conv_mm_stats_chunks). */
typedef struct vg_mm_desc {
    int32_t N, CI, CO;
    int32_t ID, IH, IW, OD, OH, OW;
    int32_t nq, ks[4];
    int32_t PDT, PH, PW, PD;
    int32_t sdi, shi, swi, d0, LD, cc;
    int32_t sdo, sho, swo, od0[4], oh0[4], ow0[4];
    int32_t relu_in, per_group, tpc, slack;
    int32_t dbuf;              /* 1: input planes double-buffered (copy of unit u+1 behind unit u's matrix work); 0: single buffer, half the LDS */
    int32_t PHB;               /* position rows per block (a block = PD planes x PHB rows x PW columns); 0 or >= PH: whole planes */
    int32_t hlo, hhi;          /* smallest / largest row offset dh in the window-offset table: which input rows a row slab stages */
    int32_t waves;             /* wavefronts per workgroup: 8 or 4 (4: half the tile, twice the co-resident workgroups per CU) */
} vg_mm_desc;
int64_t vg_conv_mm_stats_chunks(const vg_mm_desc* d, int32_t stats_per_group);
int vg_conv_mm(const vg_mm_desc* d, const float* x, const float* a_img, const int32_t* tau, const int32_t* dlt, const float* bias,
               const float* in_scale, const float* in_shift, const float* mask_src, float* y, int32_t stats_per_group,
               int32_t stats_relu, double* stats_part, void* stream);
/* dst[e] = idx[e] >= 0 ? src[idx[e]] : 0  -- builds every layer's A image from the flat fp32 parameter buffer in one launch per step */
int vg_gather_f32(const float* src, const int32_t* idx, float* dst, int64_t n, void* stream);

/* weight gradient:  dw[cb][ca][k] = sum_{n,p} PB(b)[n][cb][p] * PA(a)[n][ca][p*stride + k - pad]
 * b: [N][CB][PD][PH][PW], a: [N][CA][AD][AH][AW] (zero outside).  For a Conv3d layer b = dy,
 * a = layer input (prologue on a); for a ConvTranspose3d layer b = layer input (prologue on b),
 * a = dy.  dw comes out in the layer's own weight layout.  Replaces autograd's conv weight
 * gradients.  `ws` is caller workspace of vg_wgrad3d_ws_bytes() bytes. */
typedef struct vg_wgrad_desc {
    int32_t N, CB, CA;
    int32_t PD, PH, PW;        /* spatial size of b */
    int32_t AD, AH, AW;        /* spatial size of a */
    int32_t KD, KH, KW;
    int32_t stride;
    int32_t pad_d, pad_h, pad_w;
    int32_t pro_on_a;          /* 1: prologue applies to a, 0: to b */
    int32_t relu_in;
    int32_t per_group;
} vg_wgrad_desc;
int64_t vg_wgrad3d_ws_bytes(const vg_wgrad_desc* d);
/* accumulate != 0: dw += result (lets the caller point dw at the parameter's .grad and skip a separate add) */
int vg_wgrad3d(const vg_wgrad_desc* d, const float* a, const float* b, const float* in_scale,
               const float* in_shift, float* ws, float* dw, int32_t accumulate, void* stream);

/* The same contraction kept PER BATCH-NORM GROUP (g = n / per_group), plus one extra row: out[g][cb][ca*KVOL + tap] for cb < CB is
 * group g's share of dw; row cb == CB is the contraction of `a` with a constant-one position channel, i.e. the per-tap sums of the
 * window tensor over the positions each tap meets.  Called with in_scale = rstd, in_shift = -mean*rstd (the prologue then produces the
 * NORMALISED activation) it yields everything the batch-norm backward of the layer in front needs (vg_bn_tconv1_sums) without a pass
 * over the data.  Supported: CA == 1, 3x3x3, stride 1, no padding, CB < 16, prologue on b (the decoder's last stage); otherwise
 * VG_ERR_UNSUPPORTED.  out: float[N/per_group][CB+1][CA*KVOL], overwritten.  ws: vg_wgrad3d_grouped_ws_bytes(d) bytes. */
int64_t vg_wgrad3d_grouped_ws_bytes(const vg_wgrad_desc* d);
int vg_wgrad3d_grouped(const vg_wgrad_desc* d, const float* a, const float* b, const float* in_scale,
                       const float* in_shift, float* ws, float* out, void* stream);

/* batch-norm batch statistics (BatchNorm3d(track_running_stats=False), vae_reg_GP.py:194-196,
 * 216-218): for x [N][C][P], group g = n / per_group: mean/var over (per_group samples, P) of
 * relu?(x); writes scale = gamma*rstd, shift = beta - mean*scale, mean, rstd ([G][C] each).
 * ws: caller workspace of vg_bn_ws_bytes(N, C, P, per_group) bytes.
 * If ext_sums != NULL the kernel stops after writing the raw per-(g,c) [sum, sumsq, count]
 * triples there (double[G][C][3]) so the caller can all-reduce them across ranks, and
 * vg_bn_finalize() turns reduced triples into scale/shift/mean/rstd. */
int64_t vg_bn_ws_bytes(int32_t N, int32_t C, int64_t P, int32_t per_group);
int vg_bn_stats(const float* x, int32_t N, int32_t C, int64_t P, int32_t per_group, int32_t relu,
                const float* gamma, const float* beta, float eps, void* ws, double* ext_sums,
                float* scale, float* shift, float* mean, float* rstd, void* stream);
int vg_bn_finalize(const double* sums, int32_t G, int32_t C, const float* gamma, const float* beta,
                   float eps, float* scale, float* shift, float* mean, float* rstd, void* stream);
/* vg_bn_stats from per-block partials another kernel produced (vg_tconv3d_s2_stats): fold + finalize.
 * count = elements per (group, channel); ext_sums != NULL: write the raw [sum, sumsq, count] triples there and stop
 * (data-parallel caller all-reduces, then vg_bn_finalize); sums_ws is unused (kept for ABI stability, may be NULL). */
int vg_bn_stats_from_parts(const double* part, int32_t G, int32_t C, int64_t chunks, double count,
                           const float* gamma, const float* beta, float eps, double* ext_sums, double* sums_ws,
                           float* scale, float* shift, float* mean, float* rstd, void* stream);

/* batch-norm backward through h = relu?(p), xe = (h-mean)*rstd*gamma+beta, given dxe (in place):
 *   dgamma_part[g][c] = sum dxe*hhat, dbeta_part[g][c] = sum dxe,
 *   dp = relu'(p) * gamma*rstd * (dxe - mean(dxe) - hhat*mean(dxe*hhat)).
 * Two entry points so a data-parallel caller can all-reduce the (double[G][C][2]) sums between.
 * vg_bn_bwd_apply optionally also returns chsum[c] (+)= sum over samples and positions of dp -- the bias gradient of the
 * layer that produced p (nn.Conv3d / ConvTranspose3d bias, vae_reg_GP.py:189-215) -- from the values it already holds
 * (ws: a vg_bn_ws_bytes workspace; chsum NULL = not wanted). */
int vg_bn_bwd_reduce(const float* dxe, const float* p, int32_t N, int32_t C, int64_t P, int32_t per_group,
                     int32_t relu, const float* mean, const float* rstd, void* ws, double* sums, void* stream);
int vg_bn_bwd_apply(float* dxe_inout, const float* p, int32_t N, int32_t C, int64_t P, int32_t per_group,
                    int32_t relu, const float* gamma, const float* mean, const float* rstd,
                    const double* sums, double count, float* dgamma_part, float* dbeta_part,
                    void* ws, float* chsum, int32_t chsum_accumulate, void* stream);

/* Batch-norm backward FUSED with the data gradient of a one-output-channel 3x3x3 stride-1 ConvTranspose3d behind it -- the
 * decoder's last stage, bnt5 -> convt5 (vae_reg_GP.py:218,264), on the largest activation of the network.  The gradient that
 * reaches the batch norm, dxe[n][c][q] = sum_k dy[n][0][q + k] * w[c][0][k] (what autograd's conv_transpose3d backward / vg_corr3d
 * would write as a C-channel tensor), is recomputed from a dy tile in LDS inside both passes instead of being stored and re-read:
 *   p   [N][C][ID][IH][IW] the stored pre-activation;  dy [N][1][ID+2][IH+2][IW+2];  w [C][27] (the layer's own weight layout);
 *   reduce: sums[g][c] = {sum dxe, sum dxe*hhat}  (double[G][C][2], as vg_bn_bwd_reduce);
 *   apply : dp[N][C][...] = relu'(p) * gamma*rstd * (dxe - mean(dxe) - hhat*mean(dxe*hhat))  (as vg_bn_bwd_apply, out of place),
 *           chsum[c] (+)= per-channel sum of dp (bias gradient of the layer that produced p; NULL = not wanted).
 * ws: vg_bn_tconv1_ws_bytes(N, C, ID, per_group) bytes.  Data-parallel callers all-reduce `sums` between the two calls. */
int64_t vg_bn_tconv1_ws_bytes(int32_t N, int32_t C, int32_t ID, int32_t per_group);
int vg_bn_bwd_reduce_tconv1(const float* dy, const float* w, const float* p, int32_t N, int32_t C, int32_t ID, int32_t IH, int32_t IW,
                            int32_t per_group, int32_t relu, const float* mean, const float* rstd, void* ws, double* sums, void* stream);
int vg_bn_bwd_apply_tconv1(const float* dy, const float* w, const float* p, float* dp, int32_t N, int32_t C, int32_t ID, int32_t IH,
                           int32_t IW, int32_t per_group, int32_t relu, const float* gamma, const float* mean, const float* rstd,
                           const double* sums, double count, void* ws, float* chsum, int32_t chsum_accumulate, void* stream);

/* The reduce pass of the pair above WITHOUT reading p or dy again, from the grouped weight gradient of the transposed conv taken
 * against the normalised activation (q = vg_wgrad3d_grouped(...) with in_scale = rstd, in_shift = -mean*rstd; [G][C+1][K], K = 27):
 *   sums[g][c] = { sum_k w[c][k] q[g][C][k],  sum_k w[c][k] q[g][c][k] }         (= {sum dxe, sum dxe*hhat}, double[G][C][2])
 *   dw[c][k] (+)= gamma[c] * sum_g q[g][c][k] + beta[c] * sum_g q[g][C][k]       (the conv's own weight gradient, its layout)
 * (autograd of convt5(bnt5(.)), vae_reg_GP.py:218,264: conv_transpose3d's weight gradient and batch_norm's two reductions). */
int vg_bn_tconv1_sums(const float* q, const float* w, const float* gamma, const float* beta, int32_t G, int32_t C, int32_t K,
                      double* sums, float* dw, int32_t accumulate, void* stream);

/* per-channel sum of a [N][C][P] tensor (bias gradients): out[c] = sum_{n,p} x[n][c][p] */
int vg_channel_sum(const float* x, int32_t N, int32_t C, int64_t P, void* ws, float* out, int32_t accumulate, void* stream);

/* fused GAM accumulate + ELBO + GLM distance (vae_reg_GP.py:380,388-390,400-406).
 *   logits [G=C+1][B][V]: decoder pre-sigmoid outputs (group 0 = base map, group i = effect map i)
 *   gain   [C][B]: task_var per covariate;  x [B][V];  eps [V] float64 (log-precision map);
 *   glm [C][V] fp32 (GLM map of covariate i);
 * outputs: sum_log_prob[B] = sum_v log N(x | x_rec, exp(-eps)),  dist[C][B] = ||cons_ib - glm_i||_2,
 *   optional maps_out [G+1][B][V] = sigmoid(base), cons_1..C, full reconstruction (reconstruct path). */
int64_t vg_gam_ws_bytes(int32_t C, int32_t B, int64_t V);
int vg_gam_elbo_fwd(const float* logits, const float* gain, const float* x, const double* eps,
                    const float* glm, int32_t C, int32_t B, int64_t V, void* ws,
                    float* sum_log_prob, float* dist, float* maps_out, void* stream);
/* backward: given g_slp[B] = dL/dsum_log_prob and g_dist[C][B] = dL/ddist, writes
 *   d_logits[G][B][V] (sigmoid backward fused), d_gain[C][B], d_eps[V] (float64) and, if d_total != NULL,
 *   d_total[0] (+)= the sum of all of d_logits -- the bias gradient of the one-output-channel layer that produced the logits
 *   (convt5, vae_reg_GP.py:215,264), which then needs no pass of its own over the largest gradient tensor. */
int vg_gam_elbo_bwd(const float* logits, const float* gain, const float* x, const double* eps,
                    const float* glm, const float* dist, const float* g_slp, const float* g_dist,
                    int32_t C, int32_t B, int64_t V, void* ws,
                    float* d_logits, float* d_gain, double* d_eps, float* d_total, int32_t total_accumulate, void* stream);

/* Batch-norm parameter gradients from vg_bn_bwd_reduce's per-(group, channel) sums:
 *   dgamma[c] (+)= sum_g sums[g][c][1],  dbeta[c] (+)= sum_g sums[g][c][0]   (accumulate != 0 adds into the .grad buffers;
 *   replaces autograd's sum + add launches after torch.nn.BatchNorm3d's backward, vae_reg_GP.py:195-201,216-222). */
int vg_bn_param_grad(const double* sums, int32_t G, int32_t C, float* dgamma, float* dbeta, int32_t accumulate, void* stream);

/* Batch norm ON THE DATA (bn1, vae_reg_GP.py:187-189, 236-238) folded into conv1's backward: with dw_hat = the weight gradient taken
 * against the normalised input xhat = x*rstd + nshift (vg_data_bn_nshift: nshift = -mean*rstd) and db = the bias gradient,
 *   dw[co][ci][t] (+)= gamma[ci]*dw_hat[co][ci][t] + beta[ci]*db[co],   dbias[co] (+)= db[co],
 *   dgamma[ci] (+)= sum_{co,t} w*dw_hat,   dbeta[ci] (+)= sum_co (sum_t w[co][ci][t]) * db[co]       (accumulate != 0: add into). */
int vg_data_bn_nshift(const float* mean, const float* rstd, int32_t n, float* nshift, void* stream);
int vg_data_bn_grads(const float* dw_hat, const float* db, const float* w, const float* gamma, const float* beta,
                     int32_t CO, int32_t CI, int32_t taps, float* dw, float* dbias, float* dgamma, float* dbeta,
                     int32_t accumulate, void* stream);

/* Latent sample + KL of the low-rank Gaussian posterior in one launch (vae_reg_GP.py:321-329, 339-342, 400):
 *   d = exp(a) + 1e-6*[any(exp(a) < 1e-6)],  z = mu + w*eps_w + sqrt(d)*eps_d,
 *   kl[b] = 0.5*(-log(1 + sum w^2/d) - sum log d + sum d + sum w^2 + sum mu^2 - L),
 *   zcat[g*B + b] = [z[b], onehot_G(g)]  (the C+1 decoder inputs, row-major (G*B, L+G)).
 * mu, w, a, eps_d: [B][L]; eps_w: [B]; d_out: [B][L]; flag_out: [1] (the floor that was applied, 0 or 1e-6). */
int vg_latent_fwd(const float* mu, const float* w, const float* a, const float* eps_w, const float* eps_d,
                  int32_t B, int32_t L, int32_t G, float* zcat, float* kl, float* d_out, float* flag_out, void* stream);
/* backward of the above: g_zcat [G*B][L+G] (may be NULL), g_kl [B] (may be NULL) -> g_mu, g_w, g_a [B][L]. */
int vg_latent_bwd(const float* mu, const float* w, const float* d, const float* flag, const float* eps_w, const float* eps_d,
                  const float* g_zcat, const float* g_kl, int32_t B, int32_t L, int32_t G,
                  float* g_mu, float* g_w, float* g_a, void* stream);

/* ELBO assembly (vae_reg_GP.py:406-410): loss[0] = c_kl*sum kl[B] + c_slp*sum slp[B] + c_gp*gp_kl[0] + c_dist*sum dist[CB];
 * backward writes the four constant gradients scaled by g_loss[0]. */
int vg_loss_fwd(const float* kl, const float* slp, const float* dist, const float* gp_kl, int32_t B, int32_t CB,
                double c_kl, double c_slp, double c_gp, double c_dist, float* loss, void* stream);
int vg_loss_bwd(const float* g_loss, int32_t B, int32_t CB, double c_kl, double c_slp, double c_gp, double c_dist,
                float* g_kl, float* g_slp, float* g_dist, float* g_gp, void* stream);

/* Fully connected layers (vae_reg_GP.py:197-210 fc1..fc8, :243-259 their use; replaces torch.nn.Linear / F.relu and what autograd
 * derives from them): one strided product on the matrix cores,
 *   C[z][m][n] (+)= epilogue( sum_k A[z](m,k) * B[z](k,n) ),   m < M, n < N, k < K, z < batch,
 * A(m,k) at A + z*a_sb + m*a_sm + k*a_sk, B(k,n) at B + z*b_sb + k*b_sk + n*b_sn (element strides; one stride of each operand must
 * be 1), C rows contiguous: C + z*c_sb + m*c_sm + n.  flags:
 *   VG_FC_A_RELU  A <- max(A, 0)                       VG_FC_A_MASK  A <- A * [amask > 0]   (amask laid out as A)
 *   VG_FC_B_RELU  B <- max(B, 0)                       VG_FC_B_ONES  B gets a column n = N of ones: sum_k A(m,k) goes to cx[z*cx_sb + m]
 *   VG_FC_C_BIAS  + bias[z*bias_sb + n]                VG_FC_C_RELU  max(., 0)
 *   VG_FC_C_MASK  * [cmask > 0]  (cmask laid out as C) VG_FC_C_ACCUM add into C / cx instead of overwriting
 * ksplit > 1 (batch must be 1): the reduction is cut into ksplit pieces of ceil(K/ksplit) rounded up to 64 indices (none may be
 * empty); partial products go to ws (vg_fc_ws_bytes) and a second launch sums them in a fixed order and applies the epilogue.
 * Forward of a layer: A = x, B(k,n) = W[n][k], BIAS (+RELU).  Data gradient: A = dy (A_MASK with the layer's output if it has a ReLU),
 * B(k,n) = W[k][n].  Weight + bias gradient: A(m,k) = dy[k][m], B(k,n) = x[k][n], B_ONES, C = dW, cx = db, C_ACCUM. */
enum { VG_FC_A_RELU = 1, VG_FC_A_MASK = 2, VG_FC_B_RELU = 4, VG_FC_B_ONES = 8, VG_FC_C_BIAS = 16, VG_FC_C_RELU = 32, VG_FC_C_MASK = 64,
       VG_FC_C_ACCUM = 128 };
typedef struct vg_fc_desc {
    int32_t M, N, K, batch;
    int64_t a_sm, a_sk, a_sb;
    int64_t b_sk, b_sn, b_sb;
    int64_t c_sm, c_sb;
    int64_t bias_sb, cx_sb;
    int32_t ksplit, flags;
} vg_fc_desc;
int64_t vg_fc_ws_bytes(const vg_fc_desc* d);
int vg_fc_gemm(const vg_fc_desc* d, const float* A, const float* amask, const float* B, const float* bias, const float* cmask,
               float* C, float* cx, float* ws, void* stream);
/* Up to 4 independent products in ONE launch (a layer's data gradient and its weight + bias gradient: these products are latency-bound,
 * a launch costs as much as the arithmetic); unused pointers NULL. */
typedef struct vg_fc_job {
    vg_fc_desc d;
    const float *A, *amask, *B, *bias, *cmask;
    float *C, *cx, *ws;
} vg_fc_job;
int vg_fc_gemm_jobs(const vg_fc_job* jobs, int32_t njobs, void* stream);

/* Re-pack every conv / transposed-conv weight of the model into the [ci][tap][co] images vg_corr3d / vg_tconv3d_s2 read,
 * in one launch from the flat fp32 parameter buffer.  segs: device array [nseg][8] of int64
 * {src offset, dst offset, d0, d1, taps, mode, element count, 0} sorted by dst offset; w is [d0][d1][taps];
 * mode 0: out[i1][t][i0] = w[i0][i1][t]; mode 1: out[i0][t][i1] = w[i0][i1][t]; mode 2: as 1 with taps reversed
 * (replaces the permute/flip/contiguous copies that F.conv_transpose3d / autograd do internally). */
int vg_pack_weights(const float* flat_params, float* packed, const int64_t* segs, int32_t nseg, int64_t total, void* stream);

/* batched lower Cholesky factor of [batch][n][n] float64 SPD matrices (n <= 128), one workgroup per matrix
 * in LDS.  Replaces the Cholesky inside MultivariateNormal(beta_mean, beta_cov + 1e-5 I) (vae_reg_GP.py:368)
 * and MultivariateNormal(qu_m, qu_S) (gp.py:51); unlike hipSOLVER's potrf it can be captured into a hipGraph. */
int vg_cholesky_f64(const double* a, double* l, int32_t batch, int32_t n, void* stream);

/* The gain block of one minibatch: for every covariate i the sparse-GP posterior over the minibatch's query points, the gain
 * covariance, its B x B Cholesky factor, the reparameterised gain sample, the HRF along the batch axis and both KL terms
 * (vae_reg_GP.py:345-378 with gp.py:41-110: gp.GP.evaluate_posterior -- kernel build {Knu, Knn, Ku}, Ku solve --,
 * compute_GP_kl, calc_linW_KL, MultivariateNormal(beta_mean, beta_cov + 1e-5 I).rsample(), do_hrf_conv), forward and backward,
 * one workgroup per covariate, float64 arithmetic on fp32 inputs.
 *   table  [C][10] int64 (device): {is_gp, is_hrf, gp_index, off_sa, off_logstd, off_qu_m, off_qu_S, off_logkvar, off_log_ls, 0},
 *          offsets = element offsets of that covariate's parameters inside `params` (the flat fp32 parameter buffer);
 *   xu     [#gp][n] fp32 inducing grids (row gp_index);  covariates: element (b, i) at covariates[b*ld_cov + i], B rows (the GLOBAL
 *          batch under data parallelism);  eps_beta [C][B] fp32 standard-normal draws;  hrf_taps [hrf_taps] float64;
 *   ws     caller workspace of vg_gp_gain_ws_bytes(C, B, n) bytes: written by the forward call, read by the backward call.
 * forward outputs: task_var [C][B] fp32 (the gains), gp_kl [1] fp32 (sum over covariates of kl_lin (+ kl_gp)); optional float64
 *   copies for exports / tests (NULL = not wanted): beta_mean [C][B], beta_cov [C][B][B], f_bar [C][B], Sigma [C][B][B]
 *   (f_bar / Sigma rows of non-GP covariates are left untouched).
 * backward: given g_task_var [C][B] fp32 and g_gp_kl [1] fp32, ADDS d loss / d {sa, logstd, qu_m, qu_S, logkvar, log_ls} into
 *   flat_grads (fp32, same offsets as params).  No gradient flows to the covariates or the inducing grids (gp.py:92-101 builds
 *   the distances from Python floats). */
typedef struct vg_gain_desc {
    int32_t C, B, n;           /* covariates, minibatch, inducing points */
    int32_t hrf_taps;          /* 15 (utils.hrf(np.arange(0, 20, 1.4))); 0 = no HRF covariate */
    double jitter_b;           /* 1e-5 (vae_reg_GP.py:368) */
    double jitter_ku;          /* 0 = the reference's plain inverse of Ku (gp.py:107); > 0: Ku + jitter I (unit-variance scale) */
    double prior_var;          /* 10 (gp.py:47) */
} vg_gain_desc;
int64_t vg_gp_gain_ws_bytes(int32_t C, int32_t B, int32_t n);
int vg_gp_gain_fwd(const vg_gain_desc* d, const int64_t* table, const float* params, const float* xu,
                   const float* covariates, int64_t ld_cov, const float* eps_beta, const double* hrf_taps, void* ws,
                   float* task_var, float* gp_kl, double* beta_mean, double* beta_cov, double* f_bar, double* Sigma, void* stream);
int vg_gp_gain_bwd(const vg_gain_desc* d, const int64_t* table, const float* params, const float* xu,
                   const float* covariates, int64_t ld_cov, const float* eps_beta, const double* hrf_taps, void* ws,
                   const float* g_task_var, const float* g_gp_kl, float* flat_grads, void* stream);

/* fused Adam (torch.optim.Adam defaults, vae_reg_GP.py:179,429) over one flat buffer:
 * p,g,m,v: n elements of fp32 (is_f64 = 0) or fp64 (is_f64 = 1).  step_size = lr/(1-b1^t),
 * bc2_sqrt = sqrt(1-b2^t) are read from device scalars step_scalars[0..1] so a captured graph can replay
 * with a changing step count.
 * vg_adam_advance keeps that count ON THE DEVICE: state = double[3] {step_size, bc2_sqrt, t}; one launch does
 * t += 1 and refreshes the two scalars (torch.optim.Adam's per-step `step += 1` and bias corrections).  Launched
 * inside the step (and captured with it), so queued replays can never see a later step's scalars. */
int vg_adam_advance(double* state, double lr, double b1, double b2, void* stream);
int vg_adam_step(void* p, const void* g, void* m, void* v, int64_t n, int32_t is_f64,
                 double b1, double b2, double eps, const double* step_scalars, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VAEGAM_H */
