"""CPU oracle for the VAE-GAM train step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch (CPU, fp32) restatement of the reference algorithm for the hot
path of dannyfa/VAE-GAM (`vae_reg_GP.py`, `gp.py`, `utils.hrf`).  It exists only
to check the HIP path: only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  Nothing under `vae-gam_amd/`
imports it and the product never falls back to it.

Parity status: PINNED.  `oracle/gen_golden.py` runs the reference's own
`vae_reg_GP.VAE` (imported from /root/reference through harness-side stubs for
the absent tensorboard/nibabel/umap/torchvision modules) on seeded inputs with
injected noise and stores its outputs under `tests/golden/`; `tests/test_oracle_golden.py`
checks this restatement against those vectors.

The restatement is generalised where the reference hard-codes sizes (number of
covariates C, inducing points n, image shape) and reduces to the reference for
the cases the reference can execute (C <= 8, n = 6, 41x49x35).

Every function cites the reference lines it follows (paths relative to the
reference checkout).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

REF_NAMES = ['task', 'x', 'y', 'z', 'xrot', 'yrot', 'zrot', 'sex']  # vae_reg_GP.py:68


# --------------------------------------------------------------------------- #
# configuration / geometry
# --------------------------------------------------------------------------- #
@dataclass
class Covariate:
    name: str
    gp: bool      # continuous covariate -> linear gain + sparse-GP gain (vae_reg_GP.py:352)
    hrf: bool     # gain convolved with the HRF along the batch axis (vae_reg_GP.py:377)


def reference_schema(num_covariates: int, neural_covariates: bool = True) -> List[Covariate]:
    """Covariate roles exactly as the reference's positional rules give them.

    GP term iff 1 < i < 8 (vae_reg_GP.py:352); HRF iff neural_covariates and
    i < C-6 (vae_reg_GP.py:377); names from vae_reg_GP.py:68.  For C > 8 (no
    reference counterpart) extra continuous covariates `c1..` are
    inserted after `task`, all continuous covariates get a GP, and the HRF rule
    is kept positional.
    """
    C = num_covariates
    if C <= 8:
        names = REF_NAMES[:C]
        gp = [1 < i < 8 for i in range(1, C + 1)]
    else:
        extra = ['c%d' % k for k in range(1, C - 8 + 1)]
        names = ['task'] + extra + REF_NAMES[1:]
        gp = [1 < i < C for i in range(1, C + 1)]
    hrf = [bool(neural_covariates) and i < (C - 6) for i in range(1, C + 1)]
    return [Covariate(n, g, h) for n, g, h in zip(names, gp, hrf)]


@dataclass
class Geometry:
    """Layer geometry of `_build_network` (vae_reg_GP.py:187-218) for one image shape."""
    img: Tuple[int, int, int]
    nf: int = 8
    # decoder seed volume (channels 2*nf) and per-layer transposed-conv settings
    dec_seed: Tuple[int, int, int] = (6, 8, 5)
    convt2_pad: Tuple[int, int, int] = (1, 0, 1)
    convt2_outpad: Tuple[int, int, int] = (1, 0, 1)
    convt4_kernel: Tuple[int, int, int] = (5, 3, 3)

    @property
    def enc_flat(self) -> int:
        d, h, w = self.img
        for k, s in ((3, 1), (3, 2), (3, 1), (3, 2), (3, 1)):
            d, h, w = [(v - k) // s + 1 for v in (d, h, w)]
        return 2 * self.nf * d * h * w


def geometry_for(img: Sequence[int], nf: int = 8) -> Geometry:
    img = tuple(int(v) for v in img)
    if img == (41, 49, 35):
        return Geometry(img, nf)                                   # the reference's own network
    if img == (82, 98, 70):
        # SURVEY H1 proposed hi-res geometry (no reference counterpart):
        # dec 16x20x13 -> 18x22x15 -> (s2, no pad) 37x45x31 -> 39x47x33 -> (k4,s2) 80x96x68 -> 82x98x70
        return Geometry(img, nf, dec_seed=(16, 20, 13), convt2_pad=(0, 0, 0),
                        convt2_outpad=(0, 0, 0), convt4_kernel=(4, 4, 4))
    if img == (21, 21, 21):      # toy geometry for CPU tests (mirrors vae_gam_amd.schema)
        return Geometry(img, nf, dec_seed=(1, 1, 1), convt2_pad=(0, 0, 0), convt2_outpad=(0, 0, 0), convt4_kernel=(3, 3, 3))
    raise ValueError('no network geometry defined for image shape %r' % (img,))


@dataclass
class OracleConfig:
    num_covariates: int = 8
    num_latents: int = 32
    num_inducing_pts: int = 6
    gp_kl_scale: float = 10.0
    glm_reg_scale: float = 1.0
    neural_covariates: bool = True
    img: Tuple[int, int, int] = (41, 49, 35)
    nf: int = 8
    lr: float = 1e-3
    glm_cdist: bool = True   # True: torch.cdist as vae_reg_GP.py:388; False: the closed form B*sum_b||.||
    gp_jitter: float = 0.0   # 0: the reference's torch.inverse(Ku) (gp.py:107); > 0: Ku + jitter*k_var*I (SURVEY H2 remedy, no reference counterpart)

    @property
    def schema(self) -> List[Covariate]:
        return reference_schema(self.num_covariates, self.neural_covariates)

    @property
    def geom(self) -> Geometry:
        return geometry_for(self.img, self.nf)

    @property
    def z_dim(self) -> int:
        return self.num_latents + self.num_covariates + 1          # vae_reg_GP.py:45

    @property
    def V(self) -> int:
        return int(np.prod(self.img))


# --------------------------------------------------------------------------- #
# parameter construction (same RNG draw order as the reference's __init__)
# --------------------------------------------------------------------------- #
def _layer_shapes(cfg: OracleConfig) -> List[Tuple[str, str, Tuple[int, ...]]]:
    """(layer, kind, weight shape) in the registration order of vae_reg_GP.py:187-218."""
    nf, g = cfg.nf, cfg.geom
    zd = cfg.z_dim
    dseed = int(np.prod(g.dec_seed))
    return [
        ('conv1', 'conv', (nf, 1, 3, 3, 3)), ('conv2', 'conv', (nf, nf, 3, 3, 3)),
        ('conv3', 'conv', (2 * nf, nf, 3, 3, 3)), ('conv4', 'conv', (2 * nf, 2 * nf, 3, 3, 3)),
        ('conv5', 'conv', (2 * nf, 2 * nf, 3, 3, 3)),
        ('bn1', 'bn', (1,)), ('bn3', 'bn', (nf,)), ('bn5', 'bn', (2 * nf,)),
        ('fc1', 'fc', (200, g.enc_flat)), ('fc2', 'fc', (100, 200)),
        ('fc31', 'fc', (50, 100)), ('fc32', 'fc', (50, 100)), ('fc33', 'fc', (50, 100)),
        ('fc41', 'fc', (cfg.num_latents, 50)), ('fc42', 'fc', (cfg.num_latents, 50)),
        ('fc43', 'fc', (cfg.num_latents, 50)),
        ('fc5', 'fc', (50, zd)), ('fc6', 'fc', (100, 50)), ('fc7', 'fc', (200, 100)),
        ('fc8', 'fc', (2 * nf * dseed, 200)),
        ('convt1', 'convt', (2 * nf, 2 * nf, 3, 3, 3)), ('convt2', 'convt', (2 * nf, 2 * nf, 3, 3, 3)),
        ('convt3', 'convt', (2 * nf, nf, 3, 3, 3)), ('convt4', 'convt', (nf, nf) + tuple(g.convt4_kernel)),
        ('convt5', 'convt', (nf, 1, 3, 3, 3)),
        ('bnt1', 'bn', (2 * nf,)), ('bnt3', 'bn', (2 * nf,)), ('bnt5', 'bn', (nf,)),
    ]


def _default_init(kind: str, wshape: Tuple[int, ...]) -> Tuple[torch.Tensor, torch.Tensor]:
    """PyTorch's default layer init (what nn.Conv3d / nn.ConvTranspose3d / nn.Linear /
    nn.BatchNorm3d do at construction, vae_reg_GP.py:189-218), with the same RNG draws."""
    if kind == 'bn':
        return torch.ones(wshape), torch.zeros(wshape)
    w = torch.empty(wshape)
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    fan_in = w.size(1) * int(np.prod(wshape[2:]))      # torch's _calculate_fan_in_and_fan_out
    bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
    nbias = wshape[1] if kind == 'convt' else wshape[0]
    b = torch.empty(nbias).uniform_(-bound, bound)
    return w, b


def init_params(cfg: OracleConfig, xu_ranges: Sequence[Sequence[float]]) -> Dict[str, torch.Tensor]:
    """All trainable tensors + inducing points, drawn in the reference's order
    (vae_reg_GP.py:54-56, 72-172, 178).  Call `torch.manual_seed(seed)` first.

    `xu_ranges[k]` = [lo, hi] of the k-th GP covariate (utils.get_xu_ranges, utils.py:39-56).
    Keys: 'epsilon', 'gp.<name>.{sa,logstd,qu_m,qu_S,logkvar,log_ls,xu}', '<layer>.{weight,bias}'.
    """
    n = cfg.num_inducing_pts
    p: Dict[str, torch.Tensor] = {}
    p['epsilon'] = -math.log(10) * torch.ones(cfg.img, dtype=torch.float64)      # :54-56
    # the reference always instantiates its 8 gain-parameter sets, used or not (:68-172)
    entries = cfg.schema if cfg.num_covariates > 8 else reference_schema(8, cfg.neural_covariates)
    k = 0
    for cov in entries:
        pre = 'gp.%s.' % cov.name
        if cov.gp:
            lo, hi = xu_ranges[k]; k += 1
            p[pre + 'xu'] = torch.linspace(lo, hi, n)                            # :78
            p[pre + 'qu_m'] = torch.normal(0.0, 1.0, size=[1, n])                # :80
            p[pre + 'qu_S'] = 2 * torch.eye(n)                                   # :82
            p[pre + 'logkvar'] = torch.as_tensor(0.0)                            # :84
            p[pre + 'log_ls'] = torch.as_tensor(0.0)                             # :86
        p[pre + 'sa'] = torch.normal(1, 1, size=(1, 1))                          # :72 / :88
        p[pre + 'logstd'] = torch.normal(0, 1, size=(1, 1))                      # :74 / :90
    for name, kind, wshape in _layer_shapes(cfg):
        w, b = _default_init(kind, wshape)
        p[name + '.weight'], p[name + '.bias'] = w, b
    return p


TRAINABLE_SUFFIX_EXCLUDE = ('.xu',)


def trainable_names(params: Dict[str, torch.Tensor]) -> List[str]:
    return [k for k in params if not k.endswith(TRAINABLE_SUFFIX_EXCLUDE)]


# --------------------------------------------------------------------------- #
# network
# --------------------------------------------------------------------------- #
ReduceFn = Optional[Callable[[torch.Tensor], torch.Tensor]]


def batch_norm_stats(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                     reduce_fn: ReduceFn = None, eps: float = 1e-5) -> torch.Tensor:
    """BatchNorm3d(track_running_stats=False): batch statistics in train AND eval
    (vae_reg_GP.py:194-196, 216-218).  `reduce_fn` sums the (2, C) partial [sum, sumsq]
    across data-parallel ranks; None = single process."""
    C = x.shape[1]
    dims = (0, 2, 3, 4)
    cnt = torch.tensor(float(x.numel() // C))
    part = torch.stack([x.sum(dims), (x * x).sum(dims)])
    if reduce_fn is not None:
        part = reduce_fn(part)
        cnt = reduce_fn(cnt.clone())
    mean = part[0] / cnt
    var = part[1] / cnt - mean * mean
    sh = (1, C, 1, 1, 1)
    return (x - mean.view(sh)) * torch.rsqrt(var.view(sh) + eps) * gamma.view(sh) + beta.view(sh)


def _bn(x, p, name, reduce_fn):
    if reduce_fn is None:
        # identical call to the reference's module forward
        return F.batch_norm(x, None, None, p[name + '.weight'], p[name + '.bias'], True, 0.1, 1e-5)
    return batch_norm_stats(x, p[name + '.weight'], p[name + '.bias'], reduce_fn)


def encode(p: Dict[str, torch.Tensor], cfg: OracleConfig, x: torch.Tensor, reduce_fn: ReduceFn = None):
    """vae_reg_GP.py:236-252."""
    h = x.view(-1, 1, *cfg.img)
    h = F.relu(F.conv3d(_bn(h, p, 'bn1', reduce_fn), p['conv1.weight'], p['conv1.bias'], 1))
    h = F.relu(F.conv3d(h, p['conv2.weight'], p['conv2.bias'], 2))
    h = F.relu(F.conv3d(_bn(h, p, 'bn3', reduce_fn), p['conv3.weight'], p['conv3.bias'], 1))
    h = F.relu(F.conv3d(h, p['conv4.weight'], p['conv4.bias'], 2))
    h = F.relu(F.conv3d(_bn(h, p, 'bn5', reduce_fn), p['conv5.weight'], p['conv5.bias'], 1))
    h = h.reshape(h.shape[0], -1)
    h = F.relu(F.linear(h, p['fc1.weight'], p['fc1.bias']))
    h = F.relu(F.linear(h, p['fc2.weight'], p['fc2.bias']))
    mu = F.linear(F.relu(F.linear(h, p['fc31.weight'], p['fc31.bias'])), p['fc41.weight'], p['fc41.bias'])
    u = F.linear(F.relu(F.linear(h, p['fc32.weight'], p['fc32.bias'])), p['fc42.weight'], p['fc42.bias']).unsqueeze(-1)
    d = torch.exp(F.linear(F.relu(F.linear(h, p['fc33.weight'], p['fc33.bias'])), p['fc43.weight'], p['fc43.bias']))
    return mu, u, d


def decode(p: Dict[str, torch.Tensor], cfg: OracleConfig, z: torch.Tensor, reduce_fn: ReduceFn = None):
    """vae_reg_GP.py:254-264.  z: (B, z_dim) -> (B, V)."""
    g = cfg.geom
    h = F.relu(F.linear(z, p['fc5.weight'], p['fc5.bias']))
    h = F.relu(F.linear(h, p['fc6.weight'], p['fc6.bias']))
    h = F.relu(F.linear(h, p['fc7.weight'], p['fc7.bias']))
    h = F.relu(F.linear(h, p['fc8.weight'], p['fc8.bias']))
    h = h.view(-1, 2 * cfg.nf, *g.dec_seed)
    h = F.relu(F.conv_transpose3d(_bn(h, p, 'bnt1', reduce_fn), p['convt1.weight'], p['convt1.bias'], 1))
    h = F.relu(F.conv_transpose3d(h, p['convt2.weight'], p['convt2.bias'], 2,
                                  padding=g.convt2_pad, output_padding=g.convt2_outpad))
    h = F.relu(F.conv_transpose3d(_bn(h, p, 'bnt3', reduce_fn), p['convt3.weight'], p['convt3.bias'], 1))
    h = F.relu(F.conv_transpose3d(h, p['convt4.weight'], p['convt4.bias'], 2))
    h = F.conv_transpose3d(_bn(h, p, 'bnt5', reduce_fn), p['convt5.weight'], p['convt5.bias'], 1)
    assert tuple(h.shape[2:]) == tuple(cfg.img), (h.shape, cfg.img)
    return torch.sigmoid(h.squeeze(1).reshape(-1, cfg.V))


# --------------------------------------------------------------------------- #
# probabilistic pieces
# --------------------------------------------------------------------------- #
def hrf_kernel() -> np.ndarray:
    """utils.hrf(np.arange(0, 20, 1.4)) (utils.py:22-36, vae_reg_GP.py:292-293), float64."""
    from scipy.stats import gamma
    t = np.arange(0, 20, 1.4)
    v = gamma.pdf(t, 6) - 0.35 * gamma.pdf(t, 12)
    return v / np.max(v) * 0.6


def do_hrf_conv(task_var: torch.Tensor) -> torch.Tensor:
    """vae_reg_GP.py:283-305: causal convolution with the 15-tap HRF along the BATCH axis,
    via the (B, B+14) Toeplitz matrix the reference builds (hrf cast to fp32 on assignment)."""
    hk = torch.tensor(hrf_kernel())
    B, T = task_var.shape[0], hk.shape[0]
    shifted = torch.zeros((B, B + T - 1), dtype=task_var.dtype)
    for i in range(B):
        shifted[i, i:i + T] = hk
    out = torch.mm(task_var.unsqueeze(0), shifted)
    return out.squeeze(0)[:-(T - 1)]


def lowrank_rsample_kl(mu, u, d, eps_w, eps_d):
    """LowRankMultivariateNormal(mu,u,d).rsample() and KL(. || N(0,I))
    (vae_reg_GP.py:324-325, 400; torch.distributions.lowrank_multivariate_normal).
    eps_w: (B,1) drawn first, eps_d: (B,L)."""
    z = mu + (u @ eps_w.unsqueeze(-1)).squeeze(-1) + d.sqrt() * eps_d
    L = mu.shape[-1]
    w = u.squeeze(-1)
    cap = 1.0 + (w * w / d).sum(-1)                       # 1x1 capacitance I + W^T D^-1 W
    logdet = torch.log(cap) + torch.log(d).sum(-1)
    kl = 0.5 * (-logdet + d.sum(-1) + (w * w).sum(-1) + (mu * mu).sum(-1) - L)
    return z, kl


def lin_gain_kl(sa: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    """calc_linW_KL (vae_reg_GP.py:266-281): KL(N(sa, std^2) || N(1, 0.5^2)), torch's
    _kl_normal_normal formula."""
    var_ratio = (std / 0.5) ** 2
    t1 = ((sa - 1.0) / 0.5) ** 2
    return 0.5 * (var_ratio + t1 - 1 - var_ratio.log())


def gp_kernel(dist, k_var, ls, scale=1.0):
    """gp._distance_to_kernel (gp.py:121-136)."""
    return k_var * torch.exp(-torch.pow(scale / np.sqrt(2) / ls * dist, 2))


def gp_posterior(xu: torch.Tensor, k_var, ls, qu_m, qu_S, xq: torch.Tensor, jitter: float = 0.0):
    """gp.GP.evaluate_posterior (gp.py:67-110), vectorised.  The reference builds the
    inducing-to-query distances as arange(Xu0 - xq_j, ., step)[:n] with python floats
    (gp.py:92-94), i.e. (Xu0 - xq_j) + k*step with NO gradient to xq/Xu; Ku from the
    |i-j| striped matrix times step (gp.py:104-105); fp32 torch.inverse (gp.py:107)."""
    n = xu.shape[0]
    step = (xu[1] - xu[0]).detach()
    d0 = (xu[0].detach().double() - xq.detach().double())                 # float(...) per query point
    knu_d = (d0.unsqueeze(0) + torch.arange(n, dtype=torch.float64).unsqueeze(1) * step.double()).to(xq.dtype)
    knu = gp_kernel(knu_d, k_var, ls)                                      # (n, B)
    knn = gp_kernel(xq.unsqueeze(0) - xq.unsqueeze(1), k_var, ls)          # knn[i,:] = xq - xq[i]
    idx = torch.arange(n, dtype=xq.dtype)
    ku = gp_kernel((idx.unsqueeze(0) - idx.unsqueeze(1)).abs(), k_var, ls, step)
    if jitter:                                                             # H2 remedy: inducing prior k_var (Ku1 + jitter I)
        ku = ku + jitter * k_var * torch.eye(n, dtype=ku.dtype)
    A = knu.T @ torch.inverse(ku)
    f_bar = A @ torch.squeeze(qu_m)
    Sigma = knn + (A @ (qu_S - ku) @ A.T)
    return f_bar, Sigma


def gp_kl(qu_m, qu_S, n):
    """gp.GP.compute_GP_kl (gp.py:41-65): KL(N(qu_m, qu_S) || N(0, 10 I)) with torch's
    _kl_multivariatenormal_multivariatenormal terms (Cholesky of the unconstrained qu_S)."""
    Lp = torch.linalg.cholesky(qu_S)
    Lq_diag = math.sqrt(10.0)
    half_term1 = n * math.log(Lq_diag) - Lp.diagonal().log().sum()
    term2 = (Lp * Lp).sum() / 10.0
    term3 = (qu_m * qu_m).sum(-1) / 10.0
    return half_term1 + 0.5 * (term2 + term3 - n)          # shape (1,)


# --------------------------------------------------------------------------- #
# forward (one minibatch)
# --------------------------------------------------------------------------- #
def draw_noise(B: int, cfg: OracleConfig, generator: Optional[torch.Generator] = None):
    """The reference's draw order per forward: randn(B,1), randn(B,L), C x randn(B)."""
    g = generator
    return {'eps_w': torch.randn(B, 1, generator=g), 'eps_d': torch.randn(B, cfg.num_latents, generator=g),
            'eps_beta': torch.stack([torch.randn(B, generator=g) for _ in range(cfg.num_covariates)])}


def forward(p: Dict[str, torch.Tensor], cfg: OracleConfig, x: torch.Tensor, covariates: torch.Tensor,
            glm_maps: torch.Tensor, noise: Dict[str, torch.Tensor], reduce_fn: ReduceFn = None,
            batch_scale: Optional[Dict[str, float]] = None, keep_maps: bool = True):
    """VAE.forward (vae_reg_GP.py:307-413) without the logging / D2H side effects.

    x: (B, *img) fp32; covariates: (B, C) fp32; glm_maps: (V, C+1) float64 with column 0 the
    CSV index (vae_reg_GP.py:58-59); noise: see draw_noise.
    Returns a dict with 'loss' (shape (1,)) and every intermediate the parity tests compare.
    """
    B, C = x.shape[0], cfg.num_covariates
    dt = x.dtype                      # fp32 = the reference; fp64 (all inputs/params cast) = conditioning yardstick
    out: Dict[str, object] = {}
    mu, u, d = encode(p, cfg, x, reduce_fn)
    if bool((d < 1e-6).any()):                                           # :321-323
        d = d + 1e-6
    z, kl_z = lowrank_rsample_kl(mu, u, d, noise['eps_w'], noise['eps_d'])   # :324-325
    out.update(mu=mu, u=u, d=d, z=z, kl_z=kl_z)

    def onehot(i):
        oh = torch.zeros(B, C + 1, dtype=dt); oh[:, i] = 1.0
        return torch.cat([z, oh], 1)

    x_rec = decode(p, cfg, onehot(0), reduce_fn)                         # :326-330
    maps = {'base': x_rec}
    gp_kl_loss = torch.zeros(1, dtype=dt)
    glm_reg = torch.zeros((), dtype=dt)
    f_bars, Sigmas, task_vars, beta_means, beta_covs, gp_kls = {}, {}, {}, {}, {}, {}
    eyeB = torch.eye(B, dtype=dt)
    for i, cov in enumerate(cfg.schema, start=1):                        # :338
        diff = decode(p, cfg, onehot(i), reduce_fn)                      # :339-343
        xq = covariates[:, i - 1]
        sa = p['gp.%s.sa' % cov.name][0]
        std = p['gp.%s.logstd' % cov.name][0].exp()
        gp_kl_loss = gp_kl_loss + lin_gain_kl(sa, std)                   # :346-348
        beta_mean = sa * xq                                              # :349
        beta_cov = std.pow(2) * xq.pow(2) * eyeB                         # :350-351
        if cov.gp:                                                       # :352
            pre = 'gp.%s.' % cov.name
            kvar = p[pre + 'logkvar'].exp() + 0.1                        # :355
            ls = 3.0 * torch.sigmoid(p[pre + 'log_ls'].exp() + 0.5)      # :357
            f_bar, Sigma = gp_posterior(p[pre + 'xu'], kvar, ls, p[pre + 'qu_m'], p[pre + 'qu_S'], xq, cfg.gp_jitter)
            beta_mean = beta_mean + f_bar                                # :363
            beta_cov = beta_cov + Sigma                                  # :364
            gp_kls[cov.name] = gp_kl(p[pre + 'qu_m'], p[pre + 'qu_S'], cfg.num_inducing_pts)
            gp_kl_loss = gp_kl_loss + gp_kls[cov.name]                   # :366-367
            f_bars[cov.name], Sigmas[cov.name] = f_bar, Sigma
        Lb = torch.linalg.cholesky(beta_cov + 1e-5 * eyeB)               # :368 (MVN ctor)
        task_var = beta_mean + Lb @ noise['eps_beta'][i - 1]             # :369 rsample
        beta_means[cov.name], beta_covs[cov.name] = beta_mean, beta_cov
        if cov.hrf:                                                      # :377-378
            task_var = do_hrf_conv(task_var)
        task_vars[cov.name] = task_var
        cons = torch.einsum('b,bx->bx', task_var, diff)                  # :380
        g = glm_maps[:, i].to(dt)
        if cfg.glm_cdist:
            glm_reg = glm_reg + torch.sum(torch.cdist(cons, g.unsqueeze(0).expand(B, -1), p=2))  # :388
        else:
            nb = B if batch_scale is None else batch_scale['global_B']
            glm_reg = glm_reg + nb * torch.linalg.vector_norm(cons - g.unsqueeze(0), dim=1).sum()
        x_rec = x_rec + cons                                             # :390
        if keep_maps:
            maps[cov.name] = cons
    maps['full_rec'] = x_rec
    # ELBO (:400-408)
    scale = torch.exp(-p['epsilon'].reshape(1, -1).expand(B, -1)).to(dt)
    xf = x.reshape(B, -1)
    log_prob = -((xf - x_rec) ** 2) / (2 * scale ** 2) - scale.log() - math.log(math.sqrt(2 * math.pi))
    sum_log_prob = log_prob.sum(1)
    elbo_b = -kl_z + sum_log_prob
    if batch_scale is None:
        elbo = elbo_b.mean(0)
    else:   # data-parallel shard: local partial of the GLOBAL mean
        elbo = elbo_b.sum(0) / batch_scale['global_B']
    loss = -elbo + cfg.gp_kl_scale * gp_kl_loss + cfg.glm_reg_scale * glm_reg      # :410, shape (1,)
    out.update(loss=loss, elbo_b=elbo_b, sum_log_prob=sum_log_prob, gp_kl_loss=gp_kl_loss, glm_reg=glm_reg,
               f_bar=f_bars, Sigma=Sigmas, task_var=task_vars, beta_mean=beta_means, beta_cov=beta_covs,
               gp_kl_terms=gp_kls, maps=maps)
    return out


# --------------------------------------------------------------------------- #
# optimiser + one train step
# --------------------------------------------------------------------------- #
class AdamState:
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8) as vae_reg_GP.py:179 uses it;
    parameters that received no gradient are skipped and get no state (torch semantics)."""

    def __init__(self, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.m: Dict[str, torch.Tensor] = {}
        self.v: Dict[str, torch.Tensor] = {}
        self.t: Dict[str, int] = {}

    @torch.no_grad()
    def step(self, params: Dict[str, torch.Tensor], grads: Dict[str, Optional[torch.Tensor]]):
        for k, g in grads.items():
            if g is None:
                continue
            if k not in self.m:
                self.m[k], self.v[k], self.t[k] = torch.zeros_like(params[k]), torch.zeros_like(params[k]), 0
            self.t[k] += 1
            t = self.t[k]
            self.m[k].lerp_(g, 1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            bc1, bc2 = 1 - self.b1 ** t, 1 - self.b2 ** t
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            params[k].addcdiv_(self.m[k], denom, value=-(self.lr / bc1))


def to_float64(params, x, covariates, noise):
    """Everything cast to float64: the same algorithm evaluated without fp32 rounding, used by the
    tests to size the reference's own fp32 conditioning."""
    return ({k: v.double() for k, v in params.items()}, x.double(), covariates.double(),
            {k: v.double() for k, v in noise.items()})


def loss_and_grads(params, cfg, x, covariates, glm_maps, noise, reduce_fn=None, batch_scale=None):
    """forward + loss.backward() (vae_reg_GP.py:425-428). Returns (out, grads-by-name)."""
    names = trainable_names(params)
    leaves = {k: params[k].detach().clone().requires_grad_(True) for k in names}
    full = dict(params); full.update(leaves)
    out = forward(full, cfg, x, covariates, glm_maps, noise, reduce_fn, batch_scale)
    gl = torch.autograd.grad(out['loss'].sum(), [leaves[k] for k in names], allow_unused=True)
    return out, dict(zip(names, gl))


def train_step(params, opt: AdamState, cfg, x, covariates, glm_maps, noise, reduce_fn=None,
               batch_scale=None, grad_reduce_fn=None):
    """One iteration of train_epoch's loop body (vae_reg_GP.py:425-429); updates `params` in place."""
    out, grads = loss_and_grads(params, cfg, x, covariates, glm_maps, noise, reduce_fn, batch_scale)
    if grad_reduce_fn is not None:
        grads = {k: (None if g is None else grad_reduce_fn(g)) for k, g in grads.items()}
    opt.step(params, grads)
    return out, grads
