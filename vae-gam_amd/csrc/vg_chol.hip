// vg_chol.hip -- batched Cholesky factorisation of small SPD matrices in float64 (gfx950).
//
// The gain of every covariate is sampled as beta_mean + L eps with L = chol(beta_cov + 1e-5 I), a B x B
// matrix per covariate (vae_reg_GP.py:368-369), and the GP KL needs chol(qu_S) (gp.py:51).  hipSOLVER's
// potrf refuses stream capture on this stack, which would keep the whole train step out of a hipGraph;
// the matrices are tiny (B <= 128), so one workgroup factorises one matrix entirely in LDS.
#include "vg_common.h"
#include "../../include/vaegam.h"

namespace {

constexpr int CHOL_MAXN = 128;      // 128*128*8 B = 128 KiB of the 160 KiB LDS

// right-looking column Cholesky; a: [batch][n][n] row-major (lower triangle read), l: same shape, upper zeroed
__global__ void __launch_bounds__(256) chol_f64_k(const double* __restrict__ a, double* __restrict__ l, int n) {
    VG_DYN_SMEM(double, m);
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const double* ab = a + (size_t)b * n * n;
    double* lb = l + (size_t)b * n * n;
    for (int i = tid; i < n * n; i += nt) m[i] = ab[i];
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        const double d = sqrt(m[j * n + j]);          // NaN for a non-positive pivot, as cholesky_ex(check_errors=False)
        __syncthreads();
        for (int i = j + tid; i < n; i += nt) m[i * n + j] = (i == j) ? d : m[i * n + j] / d;
        __syncthreads();
        // trailing update of the lower triangle: m[i][k] -= m[i][j] * m[k][j],  j < k <= i < n
        const int r = n - j - 1;
        for (int t = tid; t < r * r; t += nt) {
            const int i = j + 1 + t / r, k = j + 1 + t % r;
            if (k <= i) m[i * n + k] -= m[i * n + j] * m[k * n + j];
        }
        __syncthreads();
    }
    for (int t = tid; t < n * n; t += nt) { const int i = t / n, k = t % n; lb[t] = (k <= i) ? m[t] : 0.0; }
}

}  // namespace

extern "C" int vg_cholesky_f64(const double* a, double* l, int32_t batch, int32_t n, void* stream) {
    if (!a || !l || batch <= 0 || n <= 0) { vg_set_error("vg_cholesky_f64: bad arguments"); return VG_ERR_ARG; }
    if (n > CHOL_MAXN) { vg_set_error("vg_cholesky_f64: n=%d exceeds the in-LDS limit %d", n, CHOL_MAXN); return VG_ERR_UNSUPPORTED; }
    vg_launch(chol_f64_k, dim3(batch), dim3(256), (size_t)n * n * sizeof(double), (hipStream_t)stream, a, l, (int)n);
    return vg_check_launch("cholesky_f64");
}
