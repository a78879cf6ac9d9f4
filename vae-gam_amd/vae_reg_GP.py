"""VAE-GAM model on the MI355X-native kernels -- drop-in surface of the reference's vae_reg_GP.py.

Same constructor arguments, methods, attributes, checkpoint dictionary and (seeded) initial
parameters as the reference's `VAE` (vae_reg_GP.py:35-185, 236-264, 307-539, 691-715), but the
train step runs on hand-written HIP kernels through the C ABI of include/vaegam.h:

  * the C+1 decoder calls of one forward (vae_reg_GP.py:330,343) run as ONE batched launch per
    layer over (C+1)*B samples with batch-norm statistics kept per variant;
  * activations are stored pre-activation, ReLU / batch-norm affine are applied by the consuming
    conv kernel while it stages its LDS tile, ReLU backward is fused into the data-gradient epilogue;
  * effect-map scaling, accumulation, Gaussian log-likelihood and the GLM distance are one fused
    kernel (`GamElbo`), with no (C+2)*B*V device-to-host copies unless maps are asked for;
  * the sparse-GP gains of all continuous covariates are evaluated batched, without host syncs;
  * parameters live in one flat HBM buffer updated by a fused Adam (optim.FusedAdam).

There is no CPU fallback: `forward` needs the HIP library and a GPU and says so.
"""
import datetime
import os

import numpy as np
import torch
import torch.utils.data
from torch import nn
from torch.nn import functional as F

from . import gp, ops, utils
from .optim import FusedAdam
from .schema import REF_CSV_COLS, REF_IMG_KEYS, covariate_schema, net_geometry, parameter_sets

IMG_SHAPE = (41, 49, 35)
IMG_DIM = int(np.prod(IMG_SHAPE))


class _NullWriter:
    """Stand-in when tensorboard is absent or logging is off (the reference logs images on every
    forward, vae_reg_GP.py:333-337; here that is opt-in and never inside the timed path)."""
    def __getattr__(self, name):
        return lambda *a, **k: None


class _JsonlWriter(_NullWriter):
    """TensorBoard-free observability: the scalars the reference sends to its SummaryWriter
    (utils.py:182-389, vae_reg_GP.py:703-704) as one JSON object per line in <log_dir>/scalars.jsonl, sampled per epoch.
    add_scalar / add_scalars / flush / close are real; image and figure calls are accepted and dropped."""
    def __init__(self, log_dir):
        import json
        self._json = json
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, 'scalars.jsonl')
        self._f = open(self.path, 'a')

    def add_scalar(self, tag, value, step=None, **_):
        v = float(value.detach().cpu()) if torch.is_tensor(value) else float(value)
        self._f.write(self._json.dumps({'tag': tag, 'step': None if step is None else int(step), 'value': v}) + '\n')

    def add_scalars(self, main_tag, tag_scalar_dict, step=None, **_):
        for k, v in tag_scalar_dict.items():
            self.add_scalar('%s/%s' % (main_tag, k), v, step)

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.flush(); self._f.close()
        self._f = open(self.path, 'a')            # the reference keeps using the writer after train_loop closes it


def _make_writer(log_dir, enabled):
    """tensorboard=True: the reference's SummaryWriter when the package is present; otherwise a JSONL scalar log
    when there is a directory to write to, else a null writer."""
    if enabled:
        try:
            from torch.utils.tensorboard import SummaryWriter
            return SummaryWriter(log_dir=log_dir)
        except Exception:
            pass
    if log_dir:
        try:
            return _JsonlWriter(log_dir)
        except OSError:
            pass
    return _NullWriter()


class _GpPosteriors:
    """(names, f_bar (K,B), Sigma (K,B,B)) of the GP covariates, unpacked like that tuple; the two gathers out of the gain block's
    per-covariate arrays (77 us per step for Sigma) run only when somebody asks -- exports and tests, never the train step."""
    def __init__(self, names, fb, sg, gidx):
        self.names, self._fb, self._sg, self._gidx = names, fb, sg, gidx

    def __iter__(self):
        return iter((self.names, self._fb.index_select(0, self._gidx), self._sg.index_select(0, self._gidx)))


class VAE(nn.Module):
    def __init__(self, nf=8, save_dir='', lr=1e-3, num_covariates=8, num_latents=32, device_name="auto",
                 num_inducing_pts=6, gp_kl_scale=10.0, glm_maps='', glm_reg_scale=1.0, csv_files='',
                 neural_covariates=True, *, img_shape=IMG_SHAPE, xu_ranges=None, tensorboard=False,
                 data_parallel=None, gp_jitter=0.0, dp_gain='global'):
        """Arguments up to `neural_covariates` are the reference's (vae_reg_GP.py:36-37).
        Keyword-only extensions: `img_shape` (41x49x35 or 82x98x70), `xu_ranges` (inducing-point
        ranges given directly instead of read from `csv_files`), `tensorboard` (off by default),
        `data_parallel` (a dp.DataParallelContext), `gp_jitter` (0 = the reference's plain inverse of the
        inducing-point kernel matrix Ku, gp.py:104-107; > 0: Ku + gp_jitter*I on the unit-variance scale, i.e. the
        inducing prior k_var*(Ku + jitter I), factorised by Cholesky -- needed where the inducing grid is dense against the
        length scale and Ku is singular in any precision, e.g. 64 points).  `dp_gain` (data parallel only):
        'global' (default) = the gains of the whole global minibatch are drawn jointly on every rank from the dense Bg x Bg gain
        covariance (vae_reg_GP.py:363-369; covariates all-gathered, identical seeded noise, own columns kept): exactly the one-process
        global-batch step.  Its O(Bg^3) Cholesky / triangular solves run blocked on the matrix cores' neighbours (vg_gp.hip, large-batch
        path): 0.5 + 0.7 ms at Bg = 256, 1.8 + 1.9 ms at 512, on a side stream beside the conv stacks.
        'local' = every rank draws the gains of ITS slice from that slice's own B x B block -- the block-DIAGONAL approximation of
        the joint draw: gains of volumes on different ranks are drawn independently (their covariance through the GP is dropped),
        every volume's own marginal N(beta_mean_b, Sigma_bb) is unchanged; the HRF of the neural covariates still runs along the
        GLOBAL batch index across the slices (ops.HrfAcrossRanks).  A different stochastic estimate of the loss than the 1-rank
        global-batch step (same expectation for the non-HRF covariates; for an HRF covariate the 14 volumes behind a slice boundary
        lose the cross-slice covariance terms of their convolved gain); kept as an option: its cost does not grow with the ranks.
        `glm_maps` may be a CSV path
        (reference) or an array of shape (V, C+1) whose column 0 is the CSV index column."""
        super(VAE, self).__init__()
        self.nf, self.save_dir, self.lr = nf, save_dir, lr
        self.num_covariates, self.num_latents = num_covariates, num_latents
        self.neural_covariates = neural_covariates
        self.gp_jitter = float(gp_jitter)
        self.z_dim = self.num_latents + self.num_covariates + 1
        self.img_shape = tuple(int(v) for v in img_shape)
        self.img_dim = int(np.prod(self.img_shape))
        self.geom = net_geometry(self.img_shape, nf)
        self.schema = covariate_schema(num_covariates, neural_covariates)
        self.dp = data_parallel
        assert dp_gain in ('local', 'global')
        self.dp_gain = dp_gain
        assert device_name != "cuda" or torch.cuda.is_available()
        if device_name == "auto":
            device_name = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = torch.device(device_name)           # (the reference leaves self.device unset for "cpu"/"cuda", H6)
        if self.save_dir != '' and not os.path.exists(self.save_dir):
            os.makedirs(self.save_dir)
        # log-precision map (vae_reg_GP.py:54-56)
        self.epsilon = torch.nn.Parameter(-np.log(10) * torch.ones(self.img_shape, dtype=torch.float64))
        # GLM maps (vae_reg_GP.py:58-59): (V, C+1) float64, column 0 = CSV index
        if isinstance(glm_maps, str):
            import pandas as pd
            glm_np = pd.read_csv(glm_maps).to_numpy()
        else:
            glm_np = np.asarray(glm_maps, dtype=np.float64)
        if glm_np.shape[0] != self.img_dim or glm_np.shape[1] < self.num_covariates + 1:
            raise ValueError('glm_maps has shape %r, expected (%d, >=%d)' % (glm_np.shape, self.img_dim, self.num_covariates + 1))
        self.glm_maps = torch.from_numpy(np.ascontiguousarray(glm_np)).to(self.device)
        self.glm_reg_scale = glm_reg_scale
        self.inducing_pts = num_inducing_pts
        self.gp_kl_scale = torch.as_tensor((gp_kl_scale)).to(self.device)
        self._gp_kl_scale_host = float(gp_kl_scale)          # the fused ELBO kernel takes it by value (no device read-back)
        self.max_ls = torch.as_tensor(3.0).to(self.device)
        if xu_ranges is None:
            xu_ranges = utils.get_xu_ranges(csv_files)
        # gain parameters, same RNG draw order as vae_reg_GP.py:72-172
        sets = parameter_sets(num_covariates, neural_covariates)
        self.gp_params = {c.name: {} for c in sets}
        k = 0
        for c in sets:
            d = self.gp_params[c.name]
            if c.gp:
                lo, hi = xu_ranges[k]; k += 1
                xu = torch.linspace(lo, hi, self.inducing_pts).to(self.device)
                setattr(self, 'xu_' + c.name, xu); d['xu'] = xu
                self._gain_param(c.name, 'qu_m', 'qu_m_', torch.normal(0.0, 1.0, size=[1, self.inducing_pts]))
                self._gain_param(c.name, 'qu_S', 'qu_S_', 2 * torch.eye(self.inducing_pts))
                self._gain_param(c.name, 'logkvar', 'logkvar_', torch.as_tensor((0.0)))
                self._gain_param(c.name, 'log_ls', 'logls_', torch.as_tensor((0.0)))
            self._gain_param(c.name, 'sa', 'sa_', torch.normal(1, 1, size=(1, 1)))
            self._gain_param(c.name, 'logstd', 'logstd_', torch.normal(0, 1, size=(1, 1)))
        self._build_network()
        self.to(self.device)
        named = list(self.named_parameters())
        head_groups = [['fc31.weight', 'fc32.weight', 'fc33.weight'], ['fc31.bias', 'fc32.bias', 'fc33.bias'],
                       ['fc41.weight', 'fc42.weight', 'fc43.weight'], ['fc41.bias', 'fc42.bias', 'fc43.bias']]
        self.optimizer = FusedAdam(named, lr=self.lr, contiguous_groups=head_groups)
        self._heads = None
        hv = [self.optimizer.group_views(g_) for g_ in head_groups]
        if all(v is not None for v in hv):
            H, F_, L_ = self.fc31.out_features, self.fc31.in_features, self.fc41.out_features
            shapes = [(3 * H, F_), (3 * H,), (3, L_, H), (3, 1, L_)]
            self._heads = tuple(v[0].view(sh) for v, sh in zip(hv, shapes)) + tuple(v[1].view(sh) for v, sh in zip(hv, shapes))
        # parameters that take part in training (the reference's Adam only ever creates state for those):
        # everything except the gain sets of covariates beyond num_covariates
        used_gain = {c.name for c in self.schema}
        self.optimizer.used = {i for i, (n, _) in enumerate(named) if self._gain_cov(n) in (None, *used_gain)}
        g32 = self.optimizer.groups[torch.float32]
        off32 = {self.optimizer.names[i]: g32['offs'][k] for k, i in enumerate(g32['idx'])}
        convs = [(sp.name, sp, getattr(self, sp.name).weight) for sp in self.geom.enc + self.geom.dec]
        esz, dsz = self.geom.enc_sizes(), self.geom.dec_sizes()
        sizes = {sp.name: (esz[i], esz[i + 1]) for i, sp in enumerate(self.geom.enc)}
        sizes.update({sp.name: (dsz[i], dsz[i + 1]) for i, sp in enumerate(self.geom.dec)})
        self._packed = ops.PackedWeights(convs, g32['p'], {n: off32[n + '.weight'] for n, _, _ in convs}, sizes)
        self.epoch = 0
        self.loss = {'train': {}, 'test': {}}
        ts = datetime.datetime.now().date()
        # no save_dir: nothing is written anywhere (bench, tests); with one: <save_dir>/run/<date>/ as the reference (:181-184)
        self.writer = _make_writer(os.path.join(self.save_dir, 'run', ts.strftime('%m_%d_%Y')) if self.save_dir else None, tensorboard)
        self.log_maps = False          # per-forward image logging of the reference; opt-in
        self._hrf_cache = {}
        self._gain_const_cache = {}
        self._gain_streams = {}
        self.overlap_gains = os.environ.get('VG_OVERLAP_GAINS', '1') != '0'      # run the gain algebra on a second stream beside the conv stacks
        self._glm_f32 = None
        self.use_hip_graph = False     # capture the train step into a hipGraph (bench / long runs)
        self.recon_sums = None         # per-subject map sums left by reconstruct() for build_model_recons.mk_avg_maps
        self._graphs = {}
        self.last_gp_kl = None

    # ------------------------------------------------------------------ construction helpers
    _GAIN_PREFIXES = ('sa_', 'logstd_', 'qu_m_', 'qu_S_', 'logkvar_', 'logls_')

    def _gain_cov(self, n):
        """covariate name of a gain-parameter attribute ('qu_S_xrot' -> 'xrot'), None for other parameters"""
        for pre in self._GAIN_PREFIXES:
            if n.startswith(pre):
                return n[len(pre):]
        return None

    def _gain_param(self, cov, key, attr_prefix, value):
        p = torch.nn.Parameter(value.to(self.device))
        setattr(self, attr_prefix + cov, p)
        self.gp_params[cov][key] = p

    def _build_network(self):
        """Same layers, same registration (= RNG and named_parameters) order as vae_reg_GP.py:187-218.
        The nn modules hold the parameters (state_dict / checkpoint compatibility); the arithmetic is
        done by ops.bn_conv_act on the HIP kernels."""
        nf, g = self.nf, self.geom
        e, d = g.enc, g.dec
        self.conv1 = nn.Conv3d(1, nf, 3, 1)
        self.conv2 = nn.Conv3d(nf, nf, 3, 2)
        self.conv3 = nn.Conv3d(nf, 2 * nf, 3, 1)
        self.conv4 = nn.Conv3d(2 * nf, 2 * nf, 3, 2)
        self.conv5 = nn.Conv3d(2 * nf, 2 * nf, 3, 1)
        self.bn1 = nn.BatchNorm3d(1, track_running_stats=False)
        self.bn3 = nn.BatchNorm3d(nf, track_running_stats=False)
        self.bn5 = nn.BatchNorm3d(2 * nf, track_running_stats=False)
        self.fc1 = nn.Linear(g.enc_flat, 200)
        self.fc2 = nn.Linear(200, 100)
        self.fc31 = nn.Linear(100, 50)
        self.fc32 = nn.Linear(100, 50)
        self.fc33 = nn.Linear(100, 50)
        self.fc41 = nn.Linear(50, self.num_latents)
        self.fc42 = nn.Linear(50, self.num_latents)
        self.fc43 = nn.Linear(50, self.num_latents)
        self.fc5 = nn.Linear(self.z_dim, 50)
        self.fc6 = nn.Linear(50, 100)
        self.fc7 = nn.Linear(100, 200)
        self.fc8 = nn.Linear(200, g.dec_flat)
        self.convt1 = nn.ConvTranspose3d(2 * nf, 2 * nf, d[0].k, d[0].stride)
        self.convt2 = nn.ConvTranspose3d(2 * nf, 2 * nf, d[1].k, d[1].stride, padding=d[1].pad, output_padding=d[1].outpad)
        self.convt3 = nn.ConvTranspose3d(2 * nf, nf, d[2].k, d[2].stride)
        self.convt4 = nn.ConvTranspose3d(nf, nf, d[3].k, d[3].stride)
        self.convt5 = nn.ConvTranspose3d(nf, 1, d[4].k, d[4].stride)
        self.bnt1 = nn.BatchNorm3d(2 * nf, track_running_stats=False)
        self.bnt3 = nn.BatchNorm3d(2 * nf, track_running_stats=False)
        self.bnt5 = nn.BatchNorm3d(nf, track_running_stats=False)

    def _get_layers(self):
        """name -> layer, the 28 checkpoint entries of vae_reg_GP.py:220-234."""
        names = ['fc1', 'fc2', 'fc31', 'fc32', 'fc33', 'fc41', 'fc42', 'fc43', 'fc5', 'fc6', 'fc7', 'fc8', 'bn1', 'bn3',
                 'bn5', 'bnt1', 'bnt3', 'bnt5', 'conv1', 'conv2', 'conv3', 'conv4', 'conv5', 'convt1', 'convt2', 'convt3',
                 'convt4', 'convt5']
        return {n: getattr(self, n) for n in names}

    # ------------------------------------------------------------------ network on the HIP kernels
    def _require_gpu(self, t):
        if not t.is_cuda:
            from . import _lib
            if not _lib.get_lib().host_pointers_ok:
                raise RuntimeError('vae_gam_amd runs the VAE-GAM step on HIP kernels only: tensors must be on an MI355X '
                                   '(device "cuda"); there is no CPU path.')

    def _sync(self):
        return None if self.dp is None else self.dp.bn_sync

    def _encode_pre(self, x):
        """Encoder conv stack; returns conv5's pre-activation (B, 2nf, d, h, w)."""
        self._require_gpu(x)
        e = self.geom.enc
        B = x.shape[0]
        h = x.reshape(B, 1, *self.img_shape).contiguous()
        s = self._sync()
        bca = ops.bn_conv_act
        # bias gradients of conv2 / conv4 come out of the batch-norm backward of the layers that consume them (bn3 / bn5)
        p = bca(h, self.conv1.weight, self.conv1.bias, self.bn1.weight, self.bn1.bias, e[0], False, B, True, s, self._packed)
        p = bca(p, self.conv2.weight, self.conv2.bias, None, None, e[1], True, B, False, s, self._packed, bias_grad_by_consumer=True)
        p = bca(p, self.conv3.weight, self.conv3.bias, self.bn3.weight, self.bn3.bias, e[2], True, B, False, s, self._packed,
                producer_bias=self.conv2.bias)
        p = bca(p, self.conv4.weight, self.conv4.bias, None, None, e[3], True, B, False, s, self._packed, bias_grad_by_consumer=True)
        p = bca(p, self.conv5.weight, self.conv5.bias, self.bn5.weight, self.bn5.bias, e[4], True, B, False, s, self._packed,
                producer_bias=self.conv4.bias)
        return p

    def encode(self, x):
        """x (B, *img) -> mu (B,L), u (B,L,1), d (B,L)   (vae_reg_GP.py:236-252)."""
        self._require_gpu(x)
        self._packed.refresh()
        return self._encode(x)

    def _encode_heads(self, x, stacked=False):
        """-> mu (B,L), w (B,L), a (B,L) with u = w[..., None], d = exp(a)."""
        p = self._encode_pre(x)
        la = ops.linear_act
        h = la(self.fc1, p.reshape(p.shape[0], -1), True, relu_in=True)          # conv5's ReLU (:243) in fc1's operand load
        h = la(self.fc2, h, True)
        if self._heads is not None and self._heads[0].device == h.device:
            out = ops.HeadsAct.apply(h, *self._heads)                    # the three heads: one GEMM + one batched GEMM
            return out if stacked else (out[0], out[1], out[2])
        mu = la(self.fc41, la(self.fc31, h, True), False)
        w = la(self.fc42, la(self.fc32, h, True), False)
        a = la(self.fc43, la(self.fc33, h, True), False)
        return mu, w, a

    def _encode(self, x):
        mu, w, a = self._encode_heads(x)
        return mu, w.unsqueeze(-1), torch.exp(a)

    def _decode_logits(self, z, per_group, last_bias_by_consumer=False):
        """z (N, z_dim), N = groups*per_group -> pre-sigmoid maps (N, V).  Batch-norm statistics are
        kept per group of `per_group` consecutive samples (one group per one-hot variant).
        last_bias_by_consumer: convt5's bias gradient is formed by the consumer of the logits (ops.GamElbo(..., logits_bias=))."""
        self._require_gpu(z)
        dsp = self.geom.dec
        s = self._sync()
        la = ops.linear_act
        h = la(self.fc7, la(self.fc6, la(self.fc5, z, True), True), True)
        p = la(self.fc8, h, False).view(-1, 2 * self.nf, *self.geom.dec_seed)   # ReLU applied by convt1's loader
        bca = ops.bn_conv_act
        # explicit hand-offs between neighbouring layers (ops.BnConvAct): convt2 / convt4 also accumulate the statistics of
        # bnt3 / bnt5 (st3, st5); the bias gradients of convt2 / convt4 come out of bnt3's / bnt5's backward
        p = bca(p, self.convt1.weight, self.convt1.bias, self.bnt1.weight, self.bnt1.bias, dsp[0], True, per_group, False, s, self._packed)
        p, st3 = bca(p, self.convt2.weight, self.convt2.bias, None, None, dsp[1], True, per_group, False, s, self._packed,
                     next_bn=per_group, bias_grad_by_consumer=True)
        p = bca(p, self.convt3.weight, self.convt3.bias, self.bnt3.weight, self.bnt3.bias, dsp[2], True, per_group, False, s, self._packed,
                pre_stats=st3, producer_bias=self.convt2.bias)
        p, st5 = bca(p, self.convt4.weight, self.convt4.bias, None, None, dsp[3], True, per_group, False, s, self._packed,
                     next_bn=per_group, bias_grad_by_consumer=True)
        p = bca(p, self.convt5.weight, self.convt5.bias, self.bnt5.weight, self.bnt5.bias, dsp[4], True, per_group, False, s, self._packed,
                pre_stats=st5, producer_bias=self.convt4.bias, bias_grad_by_consumer=last_bias_by_consumer)
        return p.reshape(p.shape[0], self.img_dim)

    def decode(self, z):
        """z (B, z_dim) -> sigmoid maps (B, V)   (vae_reg_GP.py:254-264)."""
        self._require_gpu(z)
        self._packed.refresh()
        return torch.sigmoid(self._decode_logits(z, z.shape[0]))

    # ------------------------------------------------------------------ probabilistic pieces
    def calc_linW_KL(self, sa, std):
        """KL(N(sa, std^2) || N(1, 0.5^2))   (vae_reg_GP.py:266-281)."""
        var_ratio = (std / 0.5).pow(2)
        t1 = ((sa - 1.0) / 0.5).pow(2)
        return 0.5 * (var_ratio + t1 - 1 - var_ratio.log())

    def _hrf_matrix(self, B, device):
        key = (B, str(device))
        if key not in self._hrf_cache:
            hk = torch.tensor(utils.hrf(np.arange(0, 20, 1.4))).float()          # fp32 on assignment, :299
            T = torch.zeros(B, B)
            for i in range(B):
                n = min(hk.shape[0], B - i)
                T[i, i:i + n] = hk[:n]
            self._hrf_cache[key] = T.to(device)
        return self._hrf_cache[key]

    def do_hrf_conv(self, covariate_vals):
        """Causal HRF convolution along the batch axis (vae_reg_GP.py:283-305): the reference's
        (B, B+14) Toeplitz product truncated to B columns, with the matrix cached per batch size."""
        B = covariate_vals.shape[0]
        return (covariate_vals.unsqueeze(0) @ self._hrf_matrix(B, covariate_vals.device)).squeeze(0)

    def draw_noise(self, B, device, out=None):
        """The reference's draws per forward, in its order: (B,1), (B,L), then C x (B,)  (SURVEY 4).
        Data-parallel: B is the GLOBAL batch and the draws come from a generator seeded identically on every
        rank, so all ranks hold the same noise and each uses its slice."""
        gen = None if self.dp is None else self.dp.noise_generator(device)
        if out is not None:
            # straight into the replayed graph's input buffers: the same generator calls in the same order produce the same values as
            # torch.randn of these shapes (one Philox stream), without three device-to-device copies in front of every replay
            assert out['eps_w'].shape == (B, 1) and out['eps_d'].shape == (B, self.num_latents) and out['eps_beta'].shape == (self.num_covariates, B)
            out['eps_w'].normal_(generator=gen); out['eps_d'].normal_(generator=gen); out['eps_beta'].normal_(generator=gen)
            return out
        return {'eps_w': torch.randn(B, 1, device=device, generator=gen),
                'eps_d': torch.randn(B, self.num_latents, device=device, generator=gen),
                'eps_beta': torch.randn(self.num_covariates, B, device=device, generator=gen)}

    def _gains(self, covariates, eps_beta, join_stream=None, hrf_in_kernel=True):
        """All C gains of a minibatch at once (vae_reg_GP.py:345-378): ONE launch, one workgroup per covariate
        (ops.GpGain -> vg_gp_gain_fwd / _bwd; float64 arithmetic).
        covariates (B, >=C) fp32 -> task_var (C, B) fp32, gp_kl_loss (1,) fp32, beta mean / covariance (float64) and the
        GP posteriors of the continuous covariates ((names, f_bar (K,B), Sigma (K,B,B)) or None)."""
        K = self._gain_consts(covariates.device, hrf_in_kernel)
        g32 = self.optimizer.groups[torch.float32]
        params = [p for n in K['names'] for p in self.gp_params[n].values() if isinstance(p, torch.nn.Parameter)]
        tv, kl, bm, bc, fb, sg, kl_terms = ops.GpGain.apply(covariates, eps_beta, K['consts'], g32['p'], g32['g'], join_stream, *params)
        self.last_gp_kl = kl_terms                      # per covariate: kl_lin (+ kl_gp), float64 (parity tests, logging)
        post = None
        if K['gidx_list']:
            post = _GpPosteriors([K['names'][i] for i in K['gidx_list']], fb, sg, K['gidx'])
        return tv, kl, bm, bc, post

    def _gains_stream(self, dev):
        if dev.type != 'cuda' or not self.overlap_gains:
            return None
        key = str(dev)
        if key not in self._gain_streams:
            self._gain_streams[key] = torch.cuda.Stream(device=dev)
        return self._gain_streams[key]

    def _gain_consts(self, dev, hrf_in_kernel=True):
        """Per device, built once: the table of gain-parameter offsets inside the flat fp32 parameter buffer, the stacked
        inducing grids and the HRF taps (fp32-rounded as the reference's Toeplitz matrix, vae_reg_GP.py:297-299)."""
        key = (str(dev), float(self.gp_jitter), int(self.inducing_pts), bool(hrf_in_kernel))   # a changed jitter / grid size builds new constants
        if key not in self._gain_const_cache:
            names = [c.name for c in self.schema]
            g32 = self.optimizer.groups[torch.float32]
            off = {self.optimizer.names[i]: g32['offs'][k] for k, i in enumerate(g32['idx'])}
            gidx = [i for i, c in enumerate(self.schema) if c.gp]
            rows = []
            for i, c in enumerate(self.schema):
                if c.gp:
                    rows.append([1, int(c.hrf and hrf_in_kernel), gidx.index(i), off['sa_' + c.name], off['logstd_' + c.name], off['qu_m_' + c.name],
                                 off['qu_S_' + c.name], off['logkvar_' + c.name], off['logls_' + c.name], 0])
                else:
                    rows.append([0, int(c.hrf and hrf_in_kernel), 0, off['sa_' + c.name], off['logstd_' + c.name], 0, 0, 0, 0, 0])
            table = torch.tensor(rows, dtype=torch.int64).to(dev)
            xu = torch.stack([self.gp_params[names[i]]['xu'].float() for i in gidx]).contiguous().to(dev) if gidx else None
            hrf = torch.tensor(utils.hrf(np.arange(0, 20, 1.4))).float().double().to(dev) if any(c.hrf for c in self.schema) \
                else torch.zeros(0, dtype=torch.float64, device=dev)
            consts = ops.GainConsts(table, xu, hrf, self.inducing_pts, jitter_ku=self.gp_jitter)
            self._gain_const_cache[key] = {'consts': consts, 'names': names, 'gidx_list': gidx,
                                           'gidx': torch.tensor(gidx, dtype=torch.int64).to(dev)}
        return self._gain_const_cache[key]

    def _glm(self):
        if self._glm_f32 is None or self._glm_f32.device != self.glm_maps.device:
            C = self.num_covariates
            self._glm_f32 = self.glm_maps[:, 1:C + 1].t().contiguous().float()              # (C, V)
        return self._glm_f32

    # ------------------------------------------------------------------ forward
    def forward_core(self, covariates, x, noise=None, want_maps=False):
        """The arithmetic of VAE.forward (vae_reg_GP.py:307-410), on device, no host syncs.
        Returns a dict of device tensors: loss (1,), z, mu, u, d, kl_z, task_var (C,B), gp_kl_loss,
        dist (C,B; glm_reg = global batch * dist.sum()), sum_log_prob, logits (C+1,B,V) and, if asked, maps (C+2,B,V)."""
        B, C, L = x.shape[0], self.num_covariates, self.num_latents
        dev = x.device
        x = x.float()
        covariates = covariates[:, :C].float()           # the loaders always carry the reference's 8 columns; covariate i reads column i-1 (:345)
        self._require_gpu(x)                         # before ANY launch: host pointers must never reach a kernel
        self._packed.refresh()                       # one launch: every conv weight -> the images the kernels read
        # data parallel: this rank holds rows [lo, lo+B) of a global batch of Bg = world*B volumes
        W, lo, Bg = 1, 0, B
        joint_gains = False
        if self.dp is not None:
            W = self.dp.world_size; lo = self.dp.rank * B; Bg = W * B
            joint_gains = self.dp_gain == 'global' and W > 1
            if joint_gains:
                covariates = self.dp.all_gather_rows(covariates)                            # (Bg, C): the gains couple the batch
        if noise is None:
            noise = self.draw_noise(Bg, dev)
        eps_w, eps_d = noise['eps_w'][lo:lo + B], noise['eps_d'][lo:lo + B]
        eps_beta = noise['eps_beta'] if (joint_gains or W == 1) else noise['eps_beta'][:, lo:lo + B].contiguous()
        # dp_gain='local': each rank draws the gains of its slice (block-diagonal approximation of the joint B x B gain covariance), but
        # the HRF still runs along the GLOBAL batch index -- the kernel then leaves the gains un-convolved and ops.HrfAcrossRanks
        # convolves them across the ranks' slices
        hrf_rows = [i for i, c in enumerate(self.schema) if c.hrf]
        hrf_across = W > 1 and not joint_gains and bool(hrf_rows)
        # The gain block (one launch forward, one backward; a single workgroup per covariate walking B serial Cholesky /
        # substitution steps) depends only on the covariates and the gain parameters: it is queued on a second HIP stream
        # beside the encoder/decoder (autograd replays its backward on that stream too, beside the decoder's backward), and
        # joined where the fused GAM/ELBO kernel needs the gains.
        gains_stream = self._gains_stream(dev)
        if gains_stream is not None:
            main = torch.cuda.current_stream(dev)
            gains_stream.wait_stream(main)                               # forks HERE: the block needs nothing the encoder produces
        heads = self._encode_heads(x, stacked=True)
        # ... but it is LAUNCHED here, between the encoder and the decoder.  In a replayed hipGraph a branch starts behind whatever its
        # parent queue held when the branch's first node was created, and autograd runs the backward nodes in reverse creation order:
        # created first, the block's forward delayed the encoder by its own length and its backward ran after everything else, alone
        # at the end of the step (rocprofv3 kernel trace: 0.10 + 0.15 ms of 7.06); created here, both sit beside the decoder's layers
        if gains_stream is not None:
            with torch.cuda.stream(gains_stream):
                gains = self._gains(covariates, eps_beta, join_stream=main, hrf_in_kernel=not hrf_across)
        else:
            gains = self._gains(covariates, eps_beta, hrf_in_kernel=not hrf_across)
        G = C + 1
        # d = exp(a) + 1e-6*[any(d < 1e-6)] (:321-323, no sync), z = rsample (:325), kl_z (:400) and the G decoder
        # inputs [z, onehot] (:326-329, 339-342) in ONE launch (and one for the backward)
        if torch.is_tensor(heads):                                       # stacked (3, B, L): one tensor in, one gradient back
            zcat, kl_z, d = ops.LatentSampleStacked.apply(heads, eps_w, eps_d, G)
            mu, w = heads[0], heads[1]
        else:
            mu, w, a = heads
            zcat, kl_z, d = ops.LatentSample.apply(mu, w, a, eps_w, eps_d, G)
        z, u = zcat[:B, :L], w.unsqueeze(-1)
        logits = self._decode_logits(zcat, B, last_bias_by_consumer=True).view(G, B, self.img_dim)
        task_var, gp_kl_loss, beta_mean, beta_cov, post = gains
        if gains_stream is not None:
            torch.cuda.current_stream(dev).wait_stream(gains_stream)
            for t in (task_var, gp_kl_loss):
                t.record_stream(torch.cuda.current_stream(dev))
        if joint_gains:
            task_var = task_var[:, lo:lo + B].contiguous()                                  # full-batch gains, this rank's columns
        if hrf_across:
            conv = ops.HrfAcrossRanks.apply(task_var[hrf_rows], self._hrf_matrix(Bg, dev), self.dp, lo)
            task_var = torch.cat([conv[hrf_rows.index(i)].unsqueeze(0) if i in hrf_rows else task_var[i:i + 1] for i in range(C)], 0)
        xf = x.reshape(B, self.img_dim)
        slp, dist = ops.GamElbo.apply(logits, task_var, xf, self.epsilon.view(-1), self._glm(), self.convt5.bias)
        # glm_reg = Bg * sum(dist) (:388-389, cdist's factor = global batch); elbo = sum(-kl_z + slp) / Bg (:406-408);
        # loss = -elbo + gp_kl_scale * gp_kl + glm_reg_scale * glm_reg (:410) -- one launch.  Replicated terms are
        # divided by the world size: the gradient all-reduce SUMS the per-rank losses.
        loss = ops.ElboLoss.apply(kl_z, slp, dist, gp_kl_loss,
                                  (1.0 / Bg, -1.0 / Bg, self._gp_kl_scale_host / W, float(self.glm_reg_scale) * Bg))
        out = dict(loss=loss, z=z, mu=mu, u=u, d=d, kl_z=kl_z, task_var=task_var, gp_kl_loss=gp_kl_loss,
                   sum_log_prob=slp, dist=dist, logits=logits, beta_mean=beta_mean, beta_cov=beta_cov,
                   gp_post=post)
        if want_maps:
            out['maps'] = ops.gam_maps(logits.detach(), task_var.detach(), xf, self.epsilon.detach().view(-1), self._glm())
        return out

    def forward(self, ids, covariates, x, log_type, return_latent_rec=False, train_mode=True, noise=None):
        """Reference signature (vae_reg_GP.py:307).  Returns the loss (shape (1,)), or
        (loss, z ndarray (B,L), imgs dict of ndarrays (B,V)) with return_latent_rec=True."""
        want_maps = return_latent_rec or (train_mode and self.log_maps)
        out = self.forward_core(covariates, x, noise, want_maps)
        if train_mode and self.log_maps:
            maps = out['maps']
            for sl in (12, 15, 18):
                utils.log_map(self.writer, self.img_shape, maps[0], sl, 'base_map', ids.shape[0], log_type)
                utils.log_map(self.writer, self.img_shape, maps[1], sl, 'task_map', ids.shape[0], log_type)
                utils.log_map(self.writer, self.img_shape, maps[-1], sl, 'full_reconstruction', ids.shape[0], log_type)
        if return_latent_rec:
            maps = out['maps'].cpu().numpy()
            imgs = {k: {} for k in REF_IMG_KEYS} if self.num_covariates <= 8 else {}
            imgs['base'] = maps[0]
            for i, c in enumerate(self.schema, start=1):
                imgs[c.img_key] = maps[i]
            imgs['full_rec'] = maps[-1]
            return out['loss'], out['z'].detach().cpu().numpy(), imgs
        return out['loss']

    # ------------------------------------------------------------------ training
    def _batch_to_device(self, sample):
        x = sample['volume'].to(self.device, non_blocking=True)
        covariates = sample['covariates'].to(self.device, non_blocking=True)
        ids = sample['subjid'].to(self.device, non_blocking=True)
        return ids, covariates, x

    def train_step(self, ids, covariates, x, noise=None):
        """One iteration of train_epoch's body (vae_reg_GP.py:425-429); returns the loss tensor (1,).
        With `self.use_hip_graph` the whole step (zero-grad, forward, backward, gradient all-reduce,
        fused Adam: a few hundred launches) is captured once per batch shape into a hipGraph and
        replayed; the host then only copies the minibatch into the graph's input buffers."""
        if self.use_hip_graph and noise is None and x.is_cuda:
            g = self._graphs.get(tuple(x.shape))
            if g is None:
                g = self._capture_step(ids, covariates, x)
            if g is not False:
                g['x'].copy_(x, non_blocking=True); g['cov'].copy_(covariates, non_blocking=True)
                Bg = x.shape[0] * (1 if self.dp is None else self.dp.world_size)
                self.draw_noise(Bg, x.device, out=g['noise'])           # same draws, same order as the eager path
                g['graph'].replay()                                     # includes the device-side Adam step-count advance
                self.optimizer.step_count += 1                          # host mirror of the device count
                return g['loss']
        return self._train_step_eager(ids, covariates, x, noise)

    def _train_step_eager(self, ids, covariates, x, noise=None):
        self.optimizer.zero_grad()
        loss = self.forward(ids, covariates, x, 'train', train_mode=True, noise=noise)
        loss.backward()
        ops.join_side_stream(x.device)                      # weight / bias gradient kernels run on a second stream
        if self.dp is not None:
            self.dp.allreduce_grads(self.optimizer.flat_grads())
        self.optimizer.advance()
        self.optimizer.apply_update()
        loss = loss.detach()
        if self.dp is not None:
            loss = self.dp.sum_scalar_tensor(loss)               # per-rank partials -> the global-batch loss
        return loss

    def _capture_step(self, ids, covariates, x):
        """Capture the train step for this batch shape; falls back to eager launches (and says so) if an
        operator in the step refuses stream capture."""
        import warnings
        key = tuple(x.shape)
        st = {'x': x.clone(), 'cov': covariates.clone(), 'ids': ids.clone()}
        # parameters / optimiser state are restored after the warm-up + capture passes, so that enabling the
        # graph does not change the training trajectory (the capture itself executes nothing)
        snap = [(g, g['p'].clone(), g['m'].clone(), g['v'].clone()) for g in self.optimizer.groups.values()]
        step0 = self.optimizer.step_count
        torch.cuda.current_stream(x.device).synchronize()       # everything queued so far has used the device-side count
        rng = torch.cuda.get_rng_state(x.device)
        try:
            if self.dp is not None and self.dp._host_staging:
                raise RuntimeError('collectives staged through the host (gloo) cannot be captured')
            st['noise'] = self.draw_noise(x.shape[0] * (1 if self.dp is None else self.dp.world_size), x.device)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                       # warm-up on a side stream (allocator, lazy inits)
                for _ in range(2):
                    self._train_step_eager(st['ids'], st['cov'], st['x'], noise=st['noise'])
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                try:
                    st['loss'] = self._train_step_eager(st['ids'], st['cov'], st['x'], noise=st['noise'])
                except Exception:
                    # every stream forked into the capture must be joined before it can end, or the capture stays open and
                    # poisons every later launch of this process
                    cur = torch.cuda.current_stream(x.device)
                    for s_ in list(self._gain_streams.values()) + list(ops._SIDE.values()):
                        cur.wait_stream(s_)
                    raise
            for g, p0, m0, v0 in snap:
                g['p'].copy_(p0); g['m'].copy_(m0); g['v'].copy_(v0)
            self.optimizer.set_step_count(step0)
            torch.cuda.set_rng_state(rng, x.device)
            st['graph'] = graph
            self._graphs[key] = st
        except Exception as e:                                  # noqa: BLE001
            warnings.warn('hipGraph capture of the train step failed (%s: %s); running eager launches' % (type(e).__name__, e))
            self._graphs[key] = False
            torch.cuda.synchronize()
            for g, p0, m0, v0 in snap:
                g['p'].copy_(p0); g['m'].copy_(m0); g['v'].copy_(v0)
            self.optimizer.set_step_count(step0)
            torch.cuda.set_rng_state(rng, x.device)
        if self.dp is not None:
            # every rank must make the same choice (a graph replays its collectives, an eager rank issues them one by one)
            bad = torch.tensor([0.0 if self._graphs[key] is not False else 1.0], device=x.device)
            if float(self.dp.allreduce_sum_(bad).item()) > 0 and self._graphs[key] is not False:
                warnings.warn('hipGraph capture failed on another rank; all ranks launch eagerly')
                self._graphs[key] = False
        return self._graphs[key]

    def train_epoch(self, train_loader):
        self.train()
        total = torch.zeros((), dtype=torch.float64, device=self.device)
        for batch_idx, sample in enumerate(train_loader):
            ids, covariates, x = self._batch_to_device(sample)
            total += self.train_step(ids, covariates, x).sum().double()
        train_loss = float(total.item()) / len(train_loader.dataset)     # one sync per epoch (the reference syncs per batch, :426)
        print('Epoch: {} Average loss: {:.4f}'.format(self.epoch, train_loss))
        self.epoch += 1
        return train_loss

    def test_epoch(self, test_loader):
        self.eval()
        total = torch.zeros((), dtype=torch.float64, device=self.device)
        with torch.no_grad():
            for i, sample in enumerate(test_loader):
                ids, covariates, x = self._batch_to_device(sample)
                total += self.forward(ids, covariates, x, 'test', train_mode=False).sum().double()
        if self.dp is not None:
            total = self.dp.sum_scalar_tensor(total)           # per-rank partials of the global-batch losses
        test_loss = float(total.item()) / len(test_loader.dataset)
        print('Test loss: {:.4f}'.format(test_loss))
        return test_loss

    def train_loop(self, loaders, epochs=100, test_freq=2, save_freq=10, save_dir=''):
        """vae_reg_GP.py:691-715 (same prints, cadence and checkpoint names)."""
        print("=" * 40)
        print("Training: epochs", self.epoch, "to", self.epoch + epochs - 1)
        print("Training set:", len(loaders['Shuffled_train'].dataset))
        print("Test set:", len(loaders['test'].dataset))
        print("=" * 40)
        for epoch in range(self.epoch, self.epoch + epochs):
            loss = self.train_epoch(loaders['Shuffled_train'])
            self.loss['train'][epoch] = loss
            self.writer.add_scalar("Loss/Train", loss, self.epoch)
            self._log_gain_scalars()
            self.writer.flush()
            if (test_freq is not None) and (epoch % test_freq == 0):
                loss = self.test_epoch(loaders['test'])
                self.loss['test'][epoch] = loss
                self.writer.add_scalar("Loss/Test", loss, self.epoch)
            if (save_freq is not None) and (epoch % save_freq == 0) and (epoch > 0) and (self.dp is None or self.dp.rank == 0):
                filename = "checkpoint_" + str(epoch).zfill(3) + '.tar'
                file_path = os.path.join(save_dir, filename)
                self.save_state(file_path)                      # data parallel: replicas are identical, rank 0 writes
        self.writer.close()

    def _log_gain_scalars(self):
        """Per-epoch gain / GP hyper-parameters (what utils.log_beta / log_qkappa_plots trace per forward in the reference)."""
        if isinstance(self.writer, _NullWriter) and not isinstance(self.writer, _JsonlWriter):
            return
        with torch.no_grad():
            for c in self.schema:
                P = self.gp_params[c.name]
                self.writer.add_scalar('gain/%s/sa' % c.name, P['sa'].reshape(-1)[0], self.epoch)
                self.writer.add_scalar('gain/%s/std' % c.name, P['logstd'].reshape(-1)[0].exp(), self.epoch)
                if c.gp:
                    self.writer.add_scalar('gp/%s/k_var' % c.name, P['logkvar'].exp() + 0.1, self.epoch)
                    self.writer.add_scalar('gp/%s/ls' % c.name, 3.0 * torch.sigmoid(P['log_ls'].exp() + 0.5), self.epoch)
            self.writer.add_scalar('epsilon/mean', self.epsilon.mean(), self.epoch)

    # ------------------------------------------------------------------ checkpoints
    def save_state(self, filename):
        """Same dictionary as vae_reg_GP.py:452-471."""
        layers = self._get_layers()
        state = {}
        for layer_name in layers:
            state[layer_name] = {k: v.detach().clone() for k, v in layers[layer_name].state_dict().items()}
        state['optimizer_state'] = self.optimizer.state_dict()
        state['loss'] = self.loss
        state['z_dim'] = self.z_dim
        state['epoch'] = self.epoch
        state['lr'] = self.lr
        state['save_dir'] = self.save_dir
        state['epsilon'] = torch.nn.Parameter(self.epsilon.detach().clone())
        state['glm_reg_scale'] = self.glm_reg_scale
        state['gp_kl_scale'] = self.gp_kl_scale
        state['inducing_pts'] = self.inducing_pts
        if self.gp_jitter:                                # extra key, only when the extension is in use (the reference's loader ignores it):
            state['gp_jitter'] = self.gp_jitter          # the posterior that was trained is the one a resumed run / the export evaluates
        gp_out = {}
        for cov, d in self.gp_params.items():
            gp_out[cov] = {k: (torch.nn.Parameter(v.detach().clone()) if isinstance(v, torch.nn.Parameter) else v.clone())
                           for k, v in d.items()}
        state['gp_params'] = gp_out
        filename = os.path.join(self.save_dir, filename)
        torch.save(state, filename)

    def load_state(self, filename):
        """vae_reg_GP.py:473-539.  Values are copied INTO the live parameters, so the optimiser keeps
        updating epsilon and the gain parameters after a resume (the reference rebinds fresh tensors
        the optimiser never sees)."""
        checkpoint = torch.load(filename, map_location=self.device, weights_only=False)
        assert checkpoint['z_dim'] == self.z_dim
        layers = self._get_layers()
        with torch.no_grad():
            for layer_name in layers:
                for k, v in checkpoint[layer_name].items():
                    layers[layer_name].state_dict()[k].copy_(v)
            self.epsilon.copy_(checkpoint['epsilon'].detach().to(self.device))
            for cov, d in checkpoint['gp_params'].items():
                for key, v in d.items():
                    tgt = self.gp_params[cov][key]
                    tgt.data.copy_(v.detach().to(self.device)) if isinstance(tgt, torch.nn.Parameter) else tgt.copy_(v.to(self.device))
        self.optimizer.load_state_dict(checkpoint['optimizer_state'])
        self.loss = checkpoint['loss']
        self.epoch = checkpoint['epoch']
        self.glm_reg_scale = checkpoint['glm_reg_scale']
        self.gp_kl_scale = torch.as_tensor(checkpoint['gp_kl_scale']).to(self.device)
        self._gp_kl_scale_host = float(torch.as_tensor(checkpoint['gp_kl_scale']).cpu())
        self.inducing_pts = checkpoint['inducing_pts']
        self.gp_jitter = float(checkpoint.get('gp_jitter', self.gp_jitter))     # absent in checkpoints of the reference
        # constants derived from the state just replaced (stacked copies of the inducing grids, jitter) and the hipGraphs that
        # captured them must not survive the load
        self._gain_const_cache.clear()
        self._graphs.clear()
        self._glm_f32 = None

    # ------------------------------------------------------------------ post-hoc (reconstruction export lives in build_model_recons)
    def reconstruct(self, loader, ref_niis, save_dirs, write_volumes=True, noise=None):
        """Reference signature (vae_reg_GP.py:585-620): one `recon_<map>.nii` per volume and map under
        save_dirs[subject]/vol_<n>/, written with the geometry of that subject's reference NIfTI.
        The maps stay on the device until they are written; per-subject sums of every map are accumulated there as
        well (`self.recon_sums`), so that build_model_recons.mk_avg_maps does not have to re-read the files.
        write_volumes=False only accumulates.  `noise`: optional callable (batch index, batch size) -> the dict forward_core takes,
        to replay recorded draws (parity tests); by default every batch draws its own, as the reference does."""
        import os
        from . import nifti
        C = self.num_covariates
        keys = ['base'] + [c.img_key for c in self.schema] + ['full_rec']
        S = len(save_dirs)
        sums = {k: torch.zeros(S, self.img_dim, device=self.device, dtype=torch.float64) for k in keys}
        counts = torch.zeros(S, device=self.device, dtype=torch.float64)
        refs = {}
        with torch.no_grad():
            for bi, sample in enumerate(loader):
                ids, covariates, x = self._batch_to_device(sample)
                out = self.forward_core(covariates, x, noise=None if noise is None else noise(bi, int(x.shape[0])), want_maps=True)
                maps = out['maps']                                         # (C+2, B, V) on the device
                idl = ids.long()
                counts.index_add_(0, idl, torch.ones_like(idl, dtype=torch.float64))
                for j, k in enumerate(keys):
                    sums[k].index_add_(0, idl, maps[j].double())
                if not write_volumes:
                    continue
                host = maps.cpu().numpy()
                vol_num = [int(v) for v in sample['vol_num'].tolist()]
                subjidx = [int(v) for v in sample['subjid'].tolist()]
                for j, k in enumerate(keys):
                    for b in range(host.shape[1]):
                        si = subjidx[b]
                        vol_dir = os.path.join(save_dirs[si], 'vol_{}'.format(vol_num[b]))
                        os.makedirs(vol_dir, exist_ok=True)
                        ref = ref_niis[si] if si < len(ref_niis) else None
                        if ref not in refs:
                            refs[ref] = ref if (ref is not None and str(ref).endswith(('.nii', '.nii.gz')) and os.path.exists(str(ref))) else None
                        nifti.write_nifti1(os.path.join(vol_dir, 'recon_{}.nii'.format(k)), host[j, b].reshape(self.img_shape), refs[ref])
        self.recon_sums = (sums, counts)
        return sums, counts

    def plot_GPs(self, csv_file='', save_dir=''):
        """The CSV part of the reference's plot_GPs (vae_reg_GP.py:641-673): for every continuous covariate one
        `<epoch>_GP_<name>_full.csv` with the data set's covariate values sorted ascending and the posterior gain mean
        `sa*x + f_bar(x)` and variance `std^2 x^2 + diag(Sigma)(x)` at each of them (columns xq, mean, vars; the index column
        holds the original row numbers, as pandas writes a sorted frame).  The reference builds the N x N posterior
        covariance of all N volumes for its diagonal; gp.posterior_diag_batched gives the diagonal directly, in float64 like the
        training path.  The matplotlib figures are not produced.  Returns {name: DataFrame}."""
        import os
        import pandas as pd
        from .schema import REF_CSV_COLS
        plot_dir = os.path.join(save_dir, str(self.epoch).zfill(3) + '_GP_plots')
        os.makedirs(plot_dir, exist_ok=True)
        data = pd.read_csv(csv_file)
        gp_cov = [(i, c) for i, c in enumerate(self.schema) if c.gp]
        out = {}
        if not gp_cov:
            return out
        f64 = torch.float64
        with torch.no_grad():
            names = [c.name for _, c in gp_cov]
            # covariate i (1-based position in the schema) reads data column REF_CSV_COLS[i-1] (:642-643, 658)
            cols = []
            for i, c in gp_cov:
                col = REF_CSV_COLS[i - 1] if (i - 1) < len(REF_CSV_COLS) and REF_CSV_COLS[i - 1] in data.columns else c.name
                cols.append(data[col].to_numpy(dtype=np.float32))
            xq32 = torch.from_numpy(np.stack(cols)).to(self.device)                        # (K, N) fp32 as the data loader gives them
            xq = xq32.to(f64)
            xu = torch.stack([self.gp_params[n]['xu'] for n in names])
            kvar = torch.stack([self.gp_params[n]['logkvar'] for n in names]).to(f64).exp() + 0.1
            ls = 3.0 * torch.sigmoid(torch.stack([self.gp_params[n]['log_ls'] for n in names]).to(f64).exp() + 0.5)
            qu_m = torch.cat([self.gp_params[n]['qu_m'] for n in names]).to(f64)
            qu_S = torch.stack([self.gp_params[n]['qu_S'] for n in names]).to(f64)
            f_bar, var = gp.posterior_diag_batched(xu, kvar, ls, qu_m, qu_S, xq, jitter=self.gp_jitter)
            sa = torch.cat([self.gp_params[n]['sa'][0] for n in names]).to(f64).unsqueeze(1)
            std = torch.cat([self.gp_params[n]['logstd'][0] for n in names]).to(f64).exp().unsqueeze(1)
            mean = (sa * xq + f_bar).cpu().numpy()
            vars_ = (std.pow(2) * xq.pow(2) + var).cpu().numpy()
        for k, n in enumerate(names):
            df = pd.DataFrame({'xq': cols[k], 'mean': mean[k], 'vars': vars_[k]}).sort_values(by=['xq'])
            df.to_csv(os.path.join(plot_dir, str(self.epoch).zfill(3) + '_GP_' + n + '_full.csv'))
            out[n] = df
        return out

    def reconstruct_batch(self, ids, covariates, x):
        """Maps of one batch as ndarrays keyed like the reference's `imgs` (vae_reg_GP.py:605)."""
        with torch.no_grad():
            _, _, imgs = self.forward(ids, covariates, x, 'reconstruction', return_latent_rec=True, train_mode=False)
        return imgs


if __name__ == "__main__":
    pass
