"""1-D sparse variational GP gain regressors (drop-in surface of the reference's gp.py).

Same API as the reference (`GP(Xu, k_var, ls, qu_m, qu_S)`, `evaluate_posterior(X_q)`,
`compute_GP_kl(num_inducing_pts, i, xq, save_dir)`; reference gp.py:13-110), re-stated without
the per-query-point Python loops and `float()` host syncs of gp.py:92-101 and without the
hard-coded `.cuda()` of gp.py:115: everything is a handful of batched device ops on whatever
device the parameters live on, and `posterior_batched` evaluates all continuous covariates of a
minibatch in one shot (what `VAE.forward` uses).
"""
import math

import torch


def distance_to_kernel(dist_mat, k_var, ls, scale_factor=1.0):
    """Gaussian kernel of a (signed) distance matrix: k_var * exp(-(s*d/(sqrt(2)*ls))^2)  (gp.py:121-136)."""
    return k_var * torch.exp(-torch.pow(scale_factor / math.sqrt(2) / ls * dist_mat, 2))


def striped_matrix(n, device=None, dtype=torch.float32):
    """|i-j| matrix (gp.py:113-119), built on the requested device."""
    idx = torch.arange(n, device=device, dtype=dtype)
    return (idx.unsqueeze(0) - idx.unsqueeze(1)).abs()


def _ku1_inverse(ku1, jitter):
    """inverse of the unit-variance inducing kernel matrix: the reference's plain inverse (gp.py:107), or -- jitter > 0, the remedy for
    grids that are dense against the length scale -- of Ku1 + jitter I through its Cholesky factor, as vg_gp_gain_fwd does."""
    if not jitter:
        return torch.linalg.inv_ex(ku1, check_errors=False).inverse, ku1
    n = ku1.shape[-1]
    kuj = ku1 + float(jitter) * torch.eye(n, device=ku1.device, dtype=ku1.dtype)
    L = torch.linalg.cholesky(kuj)
    return torch.cholesky_inverse(L), kuj


def posterior_batched(xu, k_var, ls, qu_m, qu_S, xq, jitter=0.0):
    """q(f) for K independent 1-D GPs at once.

    xu (K,n) inducing grids, k_var (K,), ls (K,), qu_m (K,n), qu_S (K,n,n), xq (K,B) query points.
    Returns f_bar (K,B), Sigma (K,B,B).  Follows gp.py:88-110: distances inducing->query are
    (Xu0 - xq_j) + k*step evaluated without gradient to xq/Xu, Ku from the striped matrix times
    the grid step, A = Knu^T Ku^-1 with an fp32 inverse.
    """
    K, n = xu.shape
    step = (xu[:, 1] - xu[:, 0]).detach()
    d0 = xu[:, 0].detach().double().unsqueeze(1) - xq.detach().double()                    # (K,B)
    kidx = torch.arange(n, device=xu.device, dtype=torch.float64)
    knu_d = d0.unsqueeze(1) + kidx.view(1, n, 1) * step.double().view(K, 1, 1)                         # (K,n,B)
    knu_d = (knu_d if jitter else knu_d.float()).to(xq.dtype)          # fp32-rounded as gp.py:90; the jittered form keeps float64 distances
    kv, l_ = k_var.view(K, 1, 1), ls.view(K, 1, 1)
    one = torch.ones((), device=xu.device, dtype=xq.dtype)
    step = step.to(xq.dtype)
    # A = Knu^T Ku^-1 does not depend on the kernel variance (it cancels).  The reference lets autograd
    # differentiate both factors and sums two large cancelling terms into d/d(k_var), which in fp32 is
    # pure rounding noise at batch 32 (tests/test_model_gpu.py); building A from UNIT-variance kernels
    # keeps the forward value (same products up to one rounding) and removes that cancellation.
    knu1 = distance_to_kernel(knu_d, one, l_)
    knn1 = distance_to_kernel(xq.unsqueeze(1) - xq.unsqueeze(2), one, l_)                   # [k,i,j] = xq_j - xq_i
    ku1 = distance_to_kernel(striped_matrix(n, xu.device, xq.dtype).unsqueeze(0) * step.view(K, 1, 1), one, l_)
    ku1_inv, ku1 = _ku1_inverse(ku1, jitter)                                               # jitter: the inducing prior is k_var (Ku1 + jitter I)
    A = knu1.transpose(1, 2) @ ku1_inv                                                      # (K,B,n)
    f_bar = (A @ qu_m.unsqueeze(-1)).squeeze(-1)
    Sigma = kv * knn1 + A @ (qu_S - kv * ku1) @ A.transpose(1, 2)
    return f_bar, Sigma


def posterior_diag_batched(xu, k_var, ls, qu_m, qu_S, xq, jitter=0.0):
    """Posterior mean and VARIANCE (diagonal of Sigma only) of K independent 1-D GPs at N query points each:
    f_bar (K,N), var (K,N) = k_var + rowsum((A M) * A), M = qu_S - k_var Ku1, in O(N n^2) time and O(N n) memory.
    The full-data-set export (vae_reg_GP.py:641-673) asks the reference for the N x N covariance of ALL volumes and
    keeps its diagonal; this is that diagonal without the N x N matrix (same A as posterior_batched)."""
    K, n = xu.shape
    step = (xu[:, 1] - xu[:, 0]).detach()
    d0 = xu[:, 0].detach().double().unsqueeze(1) - xq.detach().double()
    kidx = torch.arange(n, device=xu.device, dtype=torch.float64)
    knu_d = d0.unsqueeze(1) + kidx.view(1, n, 1) * step.double().view(K, 1, 1)
    knu_d = (knu_d if jitter else knu_d.float()).to(xq.dtype)
    kv, l_ = k_var.view(K, 1, 1), ls.view(K, 1, 1)
    one = torch.ones((), device=xu.device, dtype=xq.dtype)
    step = step.to(xq.dtype)
    knu1 = distance_to_kernel(knu_d, one, l_)
    ku1 = distance_to_kernel(striped_matrix(n, xu.device, xq.dtype).unsqueeze(0) * step.view(K, 1, 1), one, l_)
    ku1_inv, ku1 = _ku1_inverse(ku1, jitter)                                               # the posterior the model TRAINED (VAE.gp_jitter)
    A = knu1.transpose(1, 2) @ ku1_inv                                                      # (K,N,n)
    f_bar = (A @ qu_m.unsqueeze(-1)).squeeze(-1)
    var = k_var.view(K, 1) + ((A @ (qu_S - kv * ku1)) * A).sum(-1)                          # k(0) = 1 for the unit kernel
    return f_bar, var


def kl_batched(qu_m, qu_S, prior_var=10.0):
    """KL(N(qu_m, qu_S) || N(0, prior_var*I)) for K GPs (gp.py:41-65), Cholesky of the unconstrained qu_S."""
    K, n = qu_m.shape
    from . import ops
    L = ops.cholesky(qu_S)
    half_term1 = 0.5 * n * math.log(prior_var) - L.diagonal(dim1=-2, dim2=-1).log().sum(-1)
    term2 = (L * L).sum((-2, -1)) / prior_var
    term3 = (qu_m * qu_m).sum(-1) / prior_var
    return half_term1 + 0.5 * (term2 + term3 - n)


class GP():
    """1-D GP with inducing points on a grid and a Gaussian kernel (reference gp.py:13-40)."""

    def __init__(self, Xu, k_var, ls, qu_m, qu_S):
        assert len(Xu) > 1
        self.device = Xu.device
        self.n = Xu.shape[0]
        self.step = Xu[1] - Xu[0]
        self.Xu, self.k_var, self.ls, self.qu_m, self.qu_S = Xu, k_var, ls, qu_m, qu_S

    def compute_GP_kl(self, num_inducing_pts=None, i=None, xq=None, save_dir=None):
        """KL term of the non-linear gain; the extra arguments of the reference (used there only to
        dump diagnostics, gp.py:48-63) are accepted and ignored.  Returns shape (1,)."""
        return kl_batched(self.qu_m.reshape(1, -1), self.qu_S.unsqueeze(0)).reshape(1)

    def evaluate_posterior(self, X_q):
        """Posterior over the query points: (f_bar (B,), Sigma (B,B))  (gp.py:67-110)."""
        f_bar, Sigma = posterior_batched(self.Xu.unsqueeze(0), torch.as_tensor(self.k_var).reshape(1),
                                         torch.as_tensor(self.ls).reshape(1), self.qu_m.reshape(1, -1),
                                         self.qu_S.unsqueeze(0), X_q.reshape(1, -1).to(self.Xu.dtype))
        return f_bar[0], Sigma[0]
