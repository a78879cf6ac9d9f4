"""Shared bodies of the kernel parity tests: each HIP operator against a plain PyTorch fp32
reference of the same maths.  Run on CPU tensors through the host build of the kernels
(tests/emu, index-arithmetic check) and on the GPU through libvaegam_hip.so (-m gpu)."""
import numpy as np
import torch
import torch.nn.functional as F

import vae_gam_amd  # noqa: F401
from vae_gam_amd import ops
from vae_gam_amd.ops import ConvSpec

# (name, spec, small input size) -- one entry per layer of vae_reg_GP.py:189-215, spatial sizes shrunk
LAYERS = [
    ('conv1', ConvSpec('conv', 1, 8, (3, 3, 3), 1), (7, 9, 8)),
    ('conv2', ConvSpec('conv', 8, 8, (3, 3, 3), 2), (9, 11, 8)),
    ('conv3', ConvSpec('conv', 8, 16, (3, 3, 3), 1), (5, 7, 6)),
    ('conv4', ConvSpec('conv', 16, 16, (3, 3, 3), 2), (7, 9, 6)),
    ('conv5', ConvSpec('conv', 16, 16, (3, 3, 3), 1), (4, 5, 3)),
    ('convt1', ConvSpec('convt', 16, 16, (3, 3, 3), 1), (3, 4, 5)),
    ('convt2', ConvSpec('convt', 16, 16, (3, 3, 3), 2, (1, 0, 1), (1, 0, 1)), (4, 5, 4)),
    ('convt3', ConvSpec('convt', 16, 8, (3, 3, 3), 1), (4, 6, 5)),
    ('convt4', ConvSpec('convt', 8, 8, (5, 3, 3), 2), (4, 5, 4)),
    ('convt5', ConvSpec('convt', 8, 1, (3, 3, 3), 1), (5, 7, 6)),
]

# wide rows (>= 12 positions per row): the geometries that take the row-walking weight-gradient kernel and whole-plane tiles
WIDE_LAYERS = [
    ('conv1_wide', ConvSpec('conv', 1, 8, (3, 3, 3), 1), (5, 9, 16)),
    ('conv2_wide', ConvSpec('conv', 8, 8, (3, 3, 3), 2), (7, 11, 27)),
    ('conv3_wide', ConvSpec('conv', 8, 16, (3, 3, 3), 1), (4, 7, 15)),
    ('convt3_wide', ConvSpec('convt', 16, 8, (3, 3, 3), 1), (3, 6, 12)),
    ('convt4_wide', ConvSpec('convt', 8, 8, (5, 3, 3), 2), (3, 5, 12)),
    ('convt4hr_wide', ConvSpec('convt', 8, 8, (4, 4, 4), 2), (3, 3, 12)),
    ('convt5_wide', ConvSpec('convt', 8, 1, (3, 3, 3), 1), (4, 7, 13)),
    # planes so large that ONE channel's planes fill the plane-staged kernel's LDS budget (one channel per chunk, as in the 41x49x35
    # network's large layers)
    ('convt5_split', ConvSpec('convt', 3, 1, (3, 3, 3), 1), (4, 36, 36)),
    ('convt4_split', ConvSpec('convt', 8, 8, (5, 3, 3), 2), (3, 16, 16)),
    # 33 positions per row (the first and the last layer of the 41x49x35 network): the weight-gradient rows as three compile-time blocks
    ('conv1_w33', ConvSpec('conv', 1, 8, (3, 3, 3), 1), (4, 5, 35)),
    ('convt5_w33', ConvSpec('convt', 8, 1, (3, 3, 3), 1), (3, 4, 33)),
    # 30 / 32 positions per row (the large layers of the 82x98x70 network): two compile-time blocks of 4 k-steps
    ('convt3_w30', ConvSpec('convt', 16, 8, (3, 3, 3), 1), (3, 4, 30)),
    ('convt4hr_w32', ConvSpec('convt', 8, 8, (4, 4, 4), 2), (3, 3, 32)),
]


def ref_layer(p, w, b, gamma, beta, spec, relu_in, per_group):
    h = F.relu(p) if relu_in else p
    if gamma is not None:
        N, C = h.shape[:2]
        G = N // per_group
        hg = h.reshape(G, per_group, C, -1)
        mean = hg.mean((1, 3), keepdim=True)
        var = hg.var((1, 3), unbiased=False, keepdim=True)
        hg = (hg - mean) / torch.sqrt(var + 1e-5) * gamma.view(1, 1, -1, 1) + beta.view(1, 1, -1, 1)
        h = hg.reshape(h.shape)
    if spec.kind == 'conv':
        return F.conv3d(h, w, b, spec.stride)
    return F.conv_transpose3d(h, w, b, spec.stride, spec.pad, spec.outpad)


def run_layer_case(dev, name, spec, isz, with_bn, relu_in, groups, input_is_data=False, seed=0, tol=2e-4):
    g = torch.Generator().manual_seed(seed)
    per_group = 2
    N = per_group * groups
    wshape = ((spec.co, spec.ci) if spec.kind == 'conv' else (spec.ci, spec.co)) + tuple(spec.k)
    p = torch.randn((N, spec.ci) + tuple(isz), generator=g)
    w = torch.randn(wshape, generator=g) * 0.2
    b = torch.randn(spec.co, generator=g) * 0.1
    gamma = (1 + 0.3 * torch.randn(spec.ci, generator=g)) if with_bn else None
    beta = (0.2 * torch.randn(spec.ci, generator=g)) if with_bn else None
    leaves = [t.clone().requires_grad_(True) if t is not None else None for t in (p, w, b, gamma, beta)]
    y_ref = ref_layer(*leaves, spec, relu_in, per_group)
    gy = torch.randn(y_ref.shape, generator=g)
    ref_grads = torch.autograd.grad(y_ref, [t for t in leaves if t is not None], gy)
    ref_grads = list(ref_grads)
    if gamma is None:
        ref_grads += [None, None]

    dleaves = [t.to(dev).clone().requires_grad_(True) if t is not None else None for t in (p, w, b, gamma, beta)]
    if input_is_data:
        dleaves[0] = dleaves[0].detach()
    y = ops.bn_conv_act(dleaves[0], dleaves[1], dleaves[2], dleaves[3], dleaves[4], spec, relu_in, per_group, input_is_data)
    assert y.shape == y_ref.shape, (y.shape, y_ref.shape)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.detach().numpy(), rtol=tol, atol=tol, err_msg=name + ' fwd')
    ins = [t for t in dleaves if t is not None and t.requires_grad]
    grads = list(torch.autograd.grad(y, ins, gy.to(dev)))
    names = ['p', 'w', 'b', 'gamma', 'beta']
    k = 0
    for i, t in enumerate(dleaves):
        if t is None or not t.requires_grad:
            continue
        got = grads[k].cpu().numpy(); k += 1
        want = ref_grads[i].numpy()
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(got, want, rtol=5 * tol, atol=5 * tol * scale, err_msg='%s d%s' % (name, names[i]))


def ref_gam(logits, gain, x, eps, glm):
    G, B, V = logits.shape
    s = torch.sigmoid(logits)
    xrec = s[0]
    dist = []
    for i in range(1, G):
        cons = gain[i - 1][:, None] * s[i]
        dist.append(torch.linalg.vector_norm(cons - glm[i - 1][None], dim=1))
        xrec = xrec + cons
    scale = torch.exp(-eps).float()[None]
    lp = -((x - xrec) ** 2) / (2 * scale ** 2) - scale.log() - np.log(np.sqrt(2 * np.pi))
    return lp.sum(1), (torch.stack(dist) if dist else torch.zeros(0, B))


def run_gam_case(dev, C, B, V, seed=0):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(C + 1, B, V, generator=g)
    gain = torch.randn(C, B, generator=g)
    x = torch.rand(B, V, generator=g)
    eps = (-np.log(10) + 0.3 * torch.randn(V, generator=g, dtype=torch.float64))
    glm = torch.rand(C, V, generator=g)
    lv = [logits.clone().requires_grad_(True), gain.clone().requires_grad_(True), x, eps.clone().requires_grad_(True), glm]
    slp_r, dist_r = ref_gam(*lv)
    g1 = torch.randn(B, generator=g); g2 = torch.randn(C, B, generator=g)
    tgt = [lv[0], lv[3]] + ([lv[1]] if C > 0 else [])
    gr = torch.autograd.grad((slp_r * g1).sum() + (dist_r * g2).sum(), tgt)
    dv = [logits.to(dev).clone().requires_grad_(True), gain.to(dev).clone().requires_grad_(True), x.to(dev), eps.to(dev).clone().requires_grad_(True), glm.to(dev)]
    lbias = torch.nn.Parameter(torch.zeros(1, device=dev))          # stands for the bias of the layer that produced the logits
    lbias.grad = torch.zeros_like(lbias)                            # the caller binds the buffer the backward adds into (the optimiser does in the model)
    slp, dist = ops.GamElbo.apply(*dv, lbias)
    np.testing.assert_allclose(slp.detach().cpu().numpy(), slp_r.detach().numpy(), rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(dist.detach().cpu().numpy(), dist_r.detach().numpy(), rtol=2e-5, atol=1e-5)
    dtgt = [dv[0], dv[3]] + ([dv[1]] if C > 0 else [])
    gd = torch.autograd.grad((slp * g1.to(dev)).sum() + (dist * g2.to(dev)).sum(), dtgt)
    # explicit hand-off: the backward left sum(d_logits) in the producing layer's bias gradient
    np.testing.assert_allclose(float(lbias.grad), float(gr[0].double().sum()), rtol=2e-4, atol=2e-4 * float(gr[0].abs().sum()) / max(gr[0].numel() ** 0.5, 1))
    for a, b_, nm in zip(gd, gr, ('d_logits', 'd_eps', 'd_gain')):
        sc = max(1.0, float(b_.abs().max()))
        np.testing.assert_allclose(a.cpu().numpy(), b_.numpy(), rtol=2e-4, atol=2e-5 * sc, err_msg=nm)
    maps = ops.gam_maps(dv[0].detach(), dv[1].detach(), dv[2], dv[3].detach(), dv[4])
    s = torch.sigmoid(logits)
    np.testing.assert_allclose(maps[0].cpu().numpy(), s[0].numpy(), atol=1e-6)
    full = s[0] + sum(gain[i][:, None] * s[i + 1] for i in range(C))
    np.testing.assert_allclose(maps[C + 1].cpu().numpy(), full.numpy(), atol=1e-5)


def ref_latent(mu, w, a, eps_w, eps_d, G):
    """The reference's operator sequence (vae_reg_GP.py:321-329, 339-342, 400) through its own distribution classes."""
    B, L = mu.shape
    d = torch.exp(a)
    if (d < 1e-6).any():
        d = d + 1e-6
    q = torch.distributions.LowRankMultivariateNormal(mu, w.unsqueeze(-1), d)
    z = mu + w * eps_w.view(B, 1) + d.sqrt() * eps_d            # rsample with the given draws
    p = torch.distributions.MultivariateNormal(torch.zeros(L), torch.eye(L))
    kl = torch.distributions.kl_divergence(q, p)
    oh = torch.eye(G).unsqueeze(1).expand(G, B, G)
    zcat = torch.cat([z.unsqueeze(0).expand(G, B, L), oh], 2).reshape(G * B, L + G)
    return zcat, kl, d


def run_latent_case(dev, B, L, G, seed=0, tiny_d=False):
    g = torch.Generator().manual_seed(seed)
    mu = torch.randn(B, L, generator=g); w = 0.5 * torch.randn(B, L, generator=g); a = 0.7 * torch.randn(B, L, generator=g) - 0.5
    if tiny_d:
        a[B // 2, L // 3] = -15.0                                 # exp(a) < 1e-6: the batch-wide floor kicks in
    eps_w = torch.randn(B, 1, generator=g); eps_d = torch.randn(B, L, generator=g)
    gz = torch.randn(G * B, L + G, generator=g); gk = torch.randn(B, generator=g)
    rv = [t.clone().requires_grad_(True) for t in (mu, w, a)]
    zc_r, kl_r, d_r = ref_latent(*rv, eps_w, eps_d, G)
    gr = torch.autograd.grad((zc_r * gz).sum() + (kl_r * gk).sum(), rv)
    dv = [t.to(dev).clone().requires_grad_(True) for t in (mu, w, a)]
    zc, kl, d = ops.LatentSample.apply(*dv, eps_w.to(dev), eps_d.to(dev), G)
    np.testing.assert_allclose(zc.detach().cpu().numpy(), zc_r.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(kl.detach().cpu().numpy(), kl_r.detach().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(d.cpu().numpy(), d_r.detach().numpy(), rtol=1e-6, atol=0)
    gd = torch.autograd.grad((zc * gz.to(dev)).sum() + (kl * gk.to(dev)).sum(), dv)
    for x_, y_, nm in zip(gd, gr, ('g_mu', 'g_w', 'g_a')):
        np.testing.assert_allclose(x_.cpu().numpy(), y_.numpy(), rtol=2e-4, atol=2e-5, err_msg=nm)


def run_loss_case(dev, B, C, seed=0):
    g = torch.Generator().manual_seed(seed)
    kl = torch.rand(B, generator=g) * 30; slp = -1e4 * torch.rand(B, generator=g); dist = torch.rand(C, B, generator=g) * 50
    gp = torch.rand(1, generator=g) * 100
    coef = (1.0 / B, -1.0 / B, 0.37, 1e-3 * B)
    rv = [t.clone().requires_grad_(True) for t in (kl, slp, dist, gp)]
    loss_r = -((-rv[0] + rv[1]).sum(0) / B) + coef[2] * rv[3] + 1e-3 * (B * rv[2].sum())       # vae_reg_GP.py:388-389, 406-410
    gr = torch.autograd.grad(loss_r.sum(), rv)
    dv = [t.to(dev).clone().requires_grad_(True) for t in (kl, slp, dist, gp)]
    loss = ops.ElboLoss.apply(*dv, coef)
    assert loss.shape == (1,)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), loss_r.detach().numpy(), rtol=1e-5)
    gd = torch.autograd.grad(loss.sum(), dv)
    for x_, y_ in zip(gd, gr):
        np.testing.assert_allclose(x_.cpu().numpy(), y_.numpy(), rtol=1e-6, atol=0)


def run_linear_case(dev, B=24, fin=40, fout=17, seed=0):
    """LinearAct with direct accumulation into existing .grad buffers == nn.Linear + relu under autograd."""
    g = torch.Generator().manual_seed(seed)
    ref = torch.nn.Linear(fin, fout); lay = torch.nn.Linear(fin, fout).to(dev)
    with torch.no_grad():
        lay.weight.copy_(ref.weight); lay.bias.copy_(ref.bias)
    x = torch.randn(B, fin, generator=g); gy = torch.randn(B, fout, generator=g)
    for relu, relu_in in ((True, False), (False, False), (True, True)):
        xr = x.clone().requires_grad_(True)
        yr = ref(torch.relu(xr) if relu_in else xr); yr = torch.relu(yr) if relu else yr
        ref.zero_grad(); (yr * gy).sum().backward()
        for prefilled in (False, True):
            xd = x.to(dev).clone().requires_grad_(True)
            lay.weight.grad = torch.ones_like(lay.weight) if prefilled else None
            lay.bias.grad = torch.ones_like(lay.bias) if prefilled else None
            y = ops.linear_act(lay, xd, relu, relu_in=relu_in)
            (y * gy.to(dev)).sum().backward()
            off = 1.0 if prefilled else 0.0
            np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(lay.weight.grad.cpu().numpy() - off, ref.weight.grad.numpy(), rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(lay.bias.grad.cpu().numpy() - off, ref.bias.grad.numpy(), rtol=1e-4, atol=1e-5)


def run_fc_gemm_case(dev, M, N, K, a_kc, b_kc, flags=(), batch=1, ksplit=None, seed=0):
    """vg_fc_gemm (the strided product behind every fully connected layer, vae_reg_GP.py:197-210) against a float64 einsum with the
    same operand options: each operand k-contiguous or m/n-contiguous, ReLU / mask on load, the ones column (bias gradient), bias, ReLU,
    output mask, accumulation, split-K, batch."""
    from vae_gam_amd import _lib
    g = torch.Generator().manual_seed(seed + 7 * M + N + K)
    fl = 0
    for f in flags:
        fl |= getattr(_lib, 'FC_' + f)
    A = torch.randn(batch, M, K, generator=g); Bm = torch.randn(batch, K, N, generator=g)
    amask = torch.randn(batch, M, K, generator=g); cmask = torch.randn(batch, M, N, generator=g)
    bias = torch.randn(batch, N, generator=g); C0 = torch.randn(batch, M, N, generator=g); cx0 = torch.randn(batch, M, generator=g)
    Ae = A.double()
    if 'A_RELU' in flags: Ae = Ae.clamp_min(0)
    if 'A_MASK' in flags: Ae = Ae * (amask > 0)
    Be = Bm.double().clamp_min(0) if 'B_RELU' in flags else Bm.double()
    want = torch.einsum('zmk,zkn->zmn', Ae, Be)
    wx = Ae.sum(2)
    if 'C_BIAS' in flags: want = want + bias.double()[:, None, :]
    if 'C_RELU' in flags: want = want.clamp_min(0)
    if 'C_MASK' in flags: want = want * (cmask > 0)
    if 'C_ACCUM' in flags: want = want + C0.double(); wx = wx + cx0.double()
    # device layouts: A as [z][m][k] (k contiguous) or [z][k][m]; B as [z][n][k] (k contiguous) or [z][k][n]
    Ad = (A if a_kc else A.transpose(1, 2)).contiguous().to(dev); Md = (amask if a_kc else amask.transpose(1, 2)).contiguous().to(dev)
    Bd = (Bm.transpose(1, 2) if b_kc else Bm).contiguous().to(dev)
    Cd = C0.clone().to(dev); cxd = cx0.clone().to(dev)
    a_str = (K, 1, M * K) if a_kc else (1, M, M * K)
    b_str = (1, K, N * K) if b_kc else (N, 1, N * K)
    ops.fc_gemm(Ad, Bd, Cd, M, N, K, a_str, b_str, (N, M * N), fl, batch=batch, bias=bias.to(dev), bias_sb=N, amask=Md, cmask=cmask.to(dev),
                cx=cxd, cx_sb=M, ksplit=ksplit)
    scale = max(1.0, float(want.abs().max()))
    np.testing.assert_allclose(Cd.cpu().double().numpy(), want.numpy(), rtol=0, atol=2e-6 * scale * max(1.0, K ** 0.5))
    if 'B_ONES' in flags:
        np.testing.assert_allclose(cxd.cpu().double().numpy(), wx.numpy(), rtol=0, atol=2e-6 * max(1.0, float(wx.abs().max())) * max(1.0, K ** 0.5))
    else:
        assert torch.equal(cxd.cpu(), cx0), 'cx touched without VG_FC_B_ONES'


FC_GEMM_CASES = [
    # M, N, K, a_kc, b_kc, flags, batch, ksplit
    (24, 17, 40, True, True, ('C_BIAS', 'C_RELU'), 1, None),                    # a layer's forward, ragged everywhere
    (64, 200, 256, True, True, ('A_RELU', 'C_BIAS', 'C_RELU'), 1, 4),           # fc1-like: pre-activation input, split-K
    (70, 33, 100, True, False, ('A_MASK', 'C_MASK'), 1, None),                  # data gradient through two ReLUs
    (130, 41, 300, True, False, ('A_MASK',), 1, 3),                             # ... with split-K (fc8's data gradient)
    (50, 37, 64, False, False, ('A_MASK', 'B_ONES', 'B_RELU', 'C_ACCUM'), 1, None),   # weight + bias gradient, accumulated
    (150, 101, 80, False, False, ('B_ONES',), 1, 2),                            # ... written, split-K, several tiles
    (64, 32, 50, True, True, ('C_BIAS',), 3, None),                             # the three heads, batched
    (32, 50, 64, False, False, ('B_ONES', 'C_ACCUM'), 3, None),                 # their weight gradients
    (5, 3, 2, False, True, (), 1, None),                                        # smaller than a tile in every direction
    (64, 40, 96, False, False, ('A_MASK', 'B_ONES', 'C_ACCUM'), 1, None),       # both operands contiguous along m / n, multiples of 4: 16-byte loads across rows
    (36, 44, 70, False, False, ('A_RELU', 'B_ONES', 'B_RELU'), 1, None),        # ... ragged tile edges and a partial last step
    (72, 36, 200, True, False, ('A_MASK', 'C_MASK'), 1, 2),                     # data gradient with the weights read across rows, split-K
    (870, 900, 20, True, False, ('A_MASK', 'C_BIAS'), 1, None),                 # >= 768 tiles of 32 x 32: the 64 x 64 kernel, ragged edges
]


def run_fused_stats_case(dev, spec, isz, groups=2, per_group=3, seed=0):
    """Transposed-conv forward that also leaves the next BatchNorm's statistics == a separate bn_stats pass over its output;
    bn_backward_'s fused per-channel sum == channel_sum of its result."""
    g = torch.Generator().manual_seed(seed)
    N = groups * per_group
    x = torch.randn(N, spec.ci, *isz, generator=g).to(dev)
    w = (0.2 * torch.randn(spec.ci, spec.co, *spec.k, generator=g)).to(dev)
    b = torch.randn(spec.co, generator=g).to(dev)
    gamma = (1 + 0.1 * torch.randn(spec.co, generator=g)).to(dev); beta = (0.1 * torch.randn(spec.co, generator=g)).to(dev)
    wf = ops.pack_weight(w, spec, 'fwd')
    y, part = ops.conv_forward(x, wf, b, spec, True, None, None, per_group, next_bn=per_group)   # explicit hand-off
    fused = [t.clone() for t in ops.bn_stats(y, gamma, beta, True, per_group, pre=part)]
    plain = ops.bn_stats(y, gamma, beta, True, per_group)
    for a_, b_, nm in zip(fused, plain, ('scale', 'shift', 'mean', 'rstd')):
        np.testing.assert_allclose(a_.cpu().numpy(), b_.cpu().numpy(), rtol=2e-5, atol=2e-6, err_msg=nm)
    # fused bias-gradient sum of the batch-norm backward
    dxe = torch.randn(y.shape, generator=g).to(dev)
    pbg = torch.full((spec.co,), 0.5, device=dev)             # the producing layer's bias.grad: the sum is ADDED to it
    ops.bn_backward_(dxe, y, gamma, plain[2], plain[3], True, per_group, producer_bias_grad=pbg)
    want = dxe.sum((0, 2, 3, 4)) + 0.5
    np.testing.assert_allclose(pbg.cpu().numpy(), want.cpu().numpy(), rtol=2e-4, atol=2e-4 * float(want.abs().max()))


def run_conv_mm_plan_case(dev, kind, force, seed=0):
    """vg_conv_mm with a PINNED tile -- waves per workgroup, PD planes x PHB rows per block, channels per chunk, single / double
    buffered input -- against torch's own convolution: the row-slab staging (one span per plane, padded LDS pitch), the 4-wave
    workgroups, the counted-vmcnt single-buffer order, the ReLU mask fetched ahead of the last matrix phase and the per-group-run
    statistics flush must all give what the whole-plane 8-wave plan gives.
    kind: 'convt3_fwd' (16->8, 3x3x3 stride-1 transposed conv as a padded correlation, prologue = ReLU + batch-norm affine),
          'convt3_bwd' (its data gradient: 8->16 correlation, ReLU mask of the producer fused into the epilogue),
          'convt4_fwd' (8->8, 5x3x3 stride-2 transposed conv: 4 output-parity classes, statistics of the next batch norm)."""
    g = torch.Generator().manual_seed(seed)
    per_group, groups = 3, 2
    N = per_group * groups
    if kind == 'convt4_fwd':
        spec = ops.ConvSpec('convt', 8, 8, (5, 3, 3), 2); isz = (5, 9, 6)
    else:
        spec = ops.ConvSpec('convt', 16, 8, (3, 3, 3), 1); isz = (5, 9, 6)
    osz = spec.out_size(isz)
    x = torch.randn((N, spec.ci) + isz, generator=g)
    w = 0.2 * torch.randn((spec.ci, spec.co) + tuple(spec.k), generator=g)
    b = 0.1 * torch.randn(spec.co, generator=g)
    sc = 1 + 0.3 * torch.randn(groups * spec.ci, generator=g); sh = 0.2 * torch.randn(groups * spec.ci, generator=g)
    if kind.endswith('_fwd'):
        h = torch.relu(x) * sc.view(groups, 1, spec.ci, 1, 1, 1).expand(groups, per_group, spec.ci, 1, 1, 1).reshape(N, spec.ci, 1, 1, 1) \
            + sh.view(groups, 1, spec.ci, 1, 1, 1).expand(groups, per_group, spec.ci, 1, 1, 1).reshape(N, spec.ci, 1, 1, 1)
        want = F.conv_transpose3d(h, w, b, spec.stride)
        plan = ops.mm_plan(spec, 'fwd', isz, None, force=force)
        assert plan is not None, force
        nb = per_group if kind == 'convt4_fwd' else None
        got = ops.conv_mm(x.to(dev), plan, plan.gather(w.to(dev)), b.to(dev), True, sc.to(dev), sh.to(dev), per_group, None, nb)
        if nb:
            got, part = got
            gamma = torch.ones(spec.co, device=dev); beta = torch.zeros(spec.co, device=dev)
            fused = ops.bn_stats(got, gamma, beta, True, per_group, pre=part)
            plain = ops.bn_stats(got, gamma, beta, True, per_group)
            for a_, b_, nm in zip(fused, plain, ('scale', 'shift', 'mean', 'rstd')):
                np.testing.assert_allclose(a_.cpu().numpy(), b_.cpu().numpy(), rtol=2e-5, atol=2e-6, err_msg='%s %r' % (nm, force))
    else:
        dy = torch.randn((N, spec.co) + osz, generator=g)
        xm = torch.randn((N, spec.ci) + isz, generator=g)              # the producer's pre-activation: ReLU mask of the data gradient
        want = F.conv3d(dy, w) * (xm > 0)                              # d/dx of conv_transpose3d(x, w) = correlation of dy with w
        plan = ops.mm_plan(spec, 'bwd', osz, isz, force=force)
        assert plan is not None, force
        got = ops.conv_mm(dy.to(dev), plan, plan.gather(w.to(dev)), None, False, None, None, 1, xm.to(dev))
    assert (plan.waves, plan.PD, plan.PHB, plan.cc, plan.dbuf) == tuple(force)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-4, err_msg='%s %r' % (kind, force))


def run_adam_case(dev, dtype, n=5000, steps=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    p0 = torch.randn(n, generator=g, dtype=dtype)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    p = p0.to(dev).clone(); m = torch.zeros_like(p); v = torch.zeros_like(p)
    for t in range(1, steps + 1):
        gr = torch.randn(n, generator=g, dtype=dtype)
        ref.grad = gr.clone(); opt.step()
        if t == 1:
            sc = torch.zeros(3, dtype=torch.float64, device=dev)
        ops.adam_advance_(sc, 1e-3, 0.9, 0.999)                   # device-side step count + bias-correction scalars
        want_sc = [1e-3 / (1 - 0.9 ** t), np.sqrt(1 - 0.999 ** t), float(t)]
        np.testing.assert_allclose(sc.cpu().numpy(), want_sc, rtol=1e-14)
        ops.adam_step_(p, gr.to(dev), m, v, 0.9, 0.999, 1e-8, sc)
    tol = 1e-6 if dtype == torch.float32 else 1e-12
    np.testing.assert_allclose(p.cpu().numpy(), ref.detach().numpy(), rtol=tol, atol=tol)


def run_cholesky_case(dev, batch=3, n=32, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = torch.randn(batch, n, n, generator=g, dtype=torch.float64)
    a = r @ r.transpose(1, 2) + 0.5 * torch.eye(n, dtype=torch.float64)
    a_ref = a.clone().requires_grad_(True)
    l_ref = torch.linalg.cholesky(a_ref)
    w = torch.randn(batch, n, n, generator=g, dtype=torch.float64)
    (ga_ref,) = torch.autograd.grad((l_ref * w).sum(), a_ref)
    a_d = a.to(dev).clone().requires_grad_(True)
    l = ops.cholesky(a_d)
    np.testing.assert_allclose(l.detach().cpu().numpy(), l_ref.detach().numpy(), rtol=1e-10, atol=1e-10)
    (ga,) = torch.autograd.grad((l * w.to(dev)).sum(), a_d)
    np.testing.assert_allclose(ga.cpu().numpy(), ga_ref.numpy(), rtol=1e-8, atol=1e-8)


def run_gain_case(dev, B=12, n=6, jitter=0.0, seed=0, kinds=('lin_hrf', 'gp', 'gp', 'gp_hrf', 'lin')):
    """vg_gp_gain_fwd / _bwd (one workgroup per covariate: GP posterior, gain covariance, B x B Cholesky, gain sample, HRF,
    both KLs) against the float64 oracle restatement of vae_reg_GP.py:345-378 / gp.py:41-110 with autograd for the gradients."""
    import math
    import vaegam_oracle as O
    g = torch.Generator().manual_seed(seed)
    C = len(kinds)
    P, table, xus, leaves = [], [], [], []

    def put(t):
        off = sum(x.numel() for x in P); P.append(t.reshape(-1).float()); return off
    for kind in kinds:
        sa, logstd = 1 + torch.randn(1, 1, generator=g), 0.3 * torch.randn(1, 1, generator=g)
        row = [int(kind.startswith('gp')), int(kind.endswith('hrf')), len(xus), put(sa), put(logstd), 0, 0, 0, 0, 0]
        lv = {'sa': sa, 'logstd': logstd}
        if kind.startswith('gp'):
            qm = torch.randn(1, n, generator=g)
            r = 0.2 * torch.randn(n, n, generator=g)
            qS = 2 * torch.eye(n) + r @ r.t()
            lk, ll = 0.3 * torch.randn((), generator=g), 0.3 * torch.randn((), generator=g)
            row[5], row[6], row[7], row[8] = put(qm), put(qS), put(lk), put(ll)
            xus.append(torch.linspace(-4.1, 6.2, n))
            lv.update(qu_m=qm, qu_S=qS, logkvar=lk, log_ls=ll, xu=xus[-1])
        table.append(row); leaves.append(lv)
    flat = torch.cat(P)
    cov = torch.randn(B, C + 2, generator=g) * 1.5
    cov[0, :] = 6.0; cov[1, :] = -4.0
    eps = torch.randn(C, B, generator=g)
    wt, wk = torch.randn(C, B, generator=g), 0.7
    # ---- float64 reference with autograd
    ref_leaves = []
    tv_ref, kl_ref, fb_ref, sg_ref = [], 0.0, {}, {}
    for i, (kind, lv) in enumerate(zip(kinds, leaves)):
        q = {k: v.float().double().clone().requires_grad_(k != 'xu') for k, v in lv.items()}
        ref_leaves.append(q)
        xq = cov[:, i].double()
        sa, std = q['sa'][0], q['logstd'][0].exp()
        kl_ref = kl_ref + O.lin_gain_kl(sa, std)
        bm = sa * xq
        bc = std.pow(2) * xq.pow(2) * torch.eye(B, dtype=torch.float64)
        if kind.startswith('gp'):
            kvar = q['logkvar'].exp() + 0.1
            ls = 3.0 * torch.sigmoid(q['log_ls'].exp() + 0.5)
            fb, Sg = O.gp_posterior(q['xu'].float(), kvar, ls, q['qu_m'], q['qu_S'], xq, jitter)
            if jitter:
                # dense grid, cond(Ku) ~ n / jitter: autograd through the oracle's inverse(k_var Ku) makes d/d k_var a difference of
                # huge cancelling terms (A does not depend on k_var at all).  Gradients are therefore taken through the same
                # posterior written with A built from unit-variance kernels -- tied to the oracle by its forward values.
                fb_o, Sg_o = fb.detach(), Sg.detach()
                fb, Sg = _posterior_unit_variance(q['xu'].float(), kvar, ls, q['qu_m'], q['qu_S'], xq, jitter)
                np.testing.assert_allclose(fb.detach().numpy(), fb_o.numpy(), atol=2e-5 * max(1.0, float(fb_o.abs().max())))
                np.testing.assert_allclose(Sg.detach().numpy(), Sg_o.numpy(), atol=2e-5 * max(1.0, float(Sg_o.abs().max())))
            bm = bm + fb; bc = bc + Sg
            kl_ref = kl_ref + O.gp_kl(q['qu_m'], q['qu_S'], n)
            fb_ref[i], sg_ref[i] = fb.detach(), Sg.detach()
        L = torch.linalg.cholesky(bc + 1e-5 * torch.eye(B, dtype=torch.float64))
        tv = bm + L @ eps[i].double()
        if kind.endswith('hrf'):
            tv = _hrf64(tv)
        tv_ref.append(tv)
    tv_ref = torch.stack(tv_ref)
    loss_ref = (tv_ref * wt.double()).sum() + wk * kl_ref.sum()
    loss_ref.backward()
    # ---- the kernel
    consts = ops.GainConsts(torch.tensor(table, dtype=torch.int64).to(dev), torch.stack(xus).float().to(dev),
                            _hrf_taps().to(dev), n, jitter_ku=jitter)
    fp = flat.to(dev).clone().requires_grad_(True)
    fg = torch.zeros_like(fp)
    tv, kl, bm_, bc_, fb_, sg_, klt = ops.GpGain.apply(cov.to(dev), eps.to(dev), consts, fp.detach(), fg, None, fp)
    np.testing.assert_allclose(tv.detach().cpu().numpy(), tv_ref.detach().numpy(), rtol=2e-6, atol=2e-6 * float(tv_ref.abs().max()))
    np.testing.assert_allclose(float(kl.detach()), float(kl_ref.sum()), rtol=1e-6)
    # (the kernel rounds the inducing-to-query distances to fp32 as the reference's fp32 Knu does, gp.py:90; the float64 oracle
    # does not: ~1e-7 relative on a distance, amplified by cond(Ku) ~ 1e2)
    for i in fb_ref:
        np.testing.assert_allclose(fb_[i].cpu().numpy(), fb_ref[i].numpy(), atol=2e-5)
        np.testing.assert_allclose(sg_[i].cpu().numpy(), sg_ref[i].numpy(), atol=2e-5)
    ((tv * wt.to(dev)).sum() + wk * kl.sum()).backward()
    got = fg.cpu().double()
    for i, (kind, q, row) in enumerate(zip(kinds, ref_leaves, table)):
        names = ['sa', 'logstd'] + (['qu_m', 'qu_S', 'logkvar', 'log_ls'] if kind.startswith('gp') else [])
        offs = {'sa': row[3], 'logstd': row[4], 'qu_m': row[5], 'qu_S': row[6], 'logkvar': row[7], 'log_ls': row[8]}
        for nm in names:
            want = q[nm].grad.reshape(-1)
            have = got[offs[nm]:offs[nm] + want.numel()]
            # d log_ls / d logkvar are cancelling sums: tiny values carry the fp32-distance noise -- one rounding (1e-7 relative, times
            # cond(Ku)) per query point, so the floor grows with the batch: measured 7e-6 absolute on a 5e-3 gradient at B = 256
            scale = max(float(want.abs().max()), 1e-2)
            grow = max(1.0, B / 32.0) if nm in ('logkvar', 'log_ls') else 1.0
            np.testing.assert_allclose(have.numpy(), want.numpy(), rtol=2e-4, atol=2e-4 * scale * grow, err_msg='cov %d (%s) d%s' % (i, kind, nm))


def _posterior_unit_variance(xu, k_var, ls, qu_m, qu_S, xq, jitter):
    """oracle.gp_posterior (gp.py:67-110) with A = Knu^T Ku^-1 formed from UNIT-variance kernels (k_var cancels in A)."""
    import vaegam_oracle as O
    n = xu.shape[0]
    step = (xu[1] - xu[0]).double()
    knu_d = (xu[0].double() - xq).unsqueeze(0) + torch.arange(n, dtype=torch.float64).unsqueeze(1) * step
    one = torch.ones((), dtype=torch.float64)
    knu1 = O.gp_kernel(knu_d, one, ls)
    knn1 = O.gp_kernel(xq.unsqueeze(0) - xq.unsqueeze(1), one, ls)
    idx = torch.arange(n, dtype=torch.float64)
    ku1 = O.gp_kernel((idx.unsqueeze(0) - idx.unsqueeze(1)).abs(), one, ls, step) + jitter * torch.eye(n, dtype=torch.float64)
    A = knu1.T @ torch.inverse(ku1)
    return A @ torch.squeeze(qu_m), k_var * knn1 + A @ (qu_S - k_var * ku1) @ A.T


def _hrf_taps():
    from vae_gam_amd import utils
    return torch.tensor(utils.hrf(np.arange(0, 20, 1.4))).float().double()


def _hrf64(tv):
    """causal HRF along the batch index with the fp32-rounded taps, float64 accumulate (vae_reg_GP.py:283-305)"""
    hk = _hrf_taps()
    B = tv.shape[0]
    T = torch.zeros(B, B, dtype=torch.float64)
    for i in range(B):
        m = min(hk.shape[0], B - i)
        T[i, i:i + m] = hk[:m]
    return tv @ T
