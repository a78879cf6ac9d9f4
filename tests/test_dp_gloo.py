"""Data-parallel path on CPU: 2 ranks (gloo) each running the PRODUCT model on half of a global minibatch
through the host build of the kernels == 1 rank on the whole minibatch: batch-norm statistics are
all-reduced per layer, the gains are evaluated on the all-gathered covariates with identical noise, the flat
gradient buffers are summed once.  Also the sharded iteration order of DeviceResidentData."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import toy_case as T
from vae_gam_amd import _lib

# 8 volumes: the toy decoder seed is 1x1x1, so its first batch-norm sees only B values per statistic; at B=4 the
# variance of 4 numbers is so ill-conditioned that a different fp32 summation order (2+2 vs 4) moves gradients by 5e-3
B_GLOBAL, C = 8, 3


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _single_process_reference():
    T.load_emu_library()
    x, cov, xu, glm = T.make_inputs(B_GLOBAL, C, seed=11)
    model = T.make_model(C, xu, glm)
    gen = torch.Generator().manual_seed(1234)
    noise = {'eps_w': torch.randn(B_GLOBAL, 1, generator=gen), 'eps_d': torch.randn(B_GLOBAL, 32, generator=gen),
             'eps_beta': torch.randn(C, B_GLOBAL, generator=gen)}
    loss = model.train_step(torch.zeros(B_GLOBAL, dtype=torch.int64), cov, x, noise=noise)
    g32 = model.optimizer.groups[torch.float32]
    return float(loss), g32['g'].clone(), g32['p'].clone(), model.epsilon.detach().clone()


def _rank_main(rank, world, port, out_dir, dp_gain='global', C=C):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from vae_gam_amd import dp as dpmod
    T.load_emu_library()
    ctx = dpmod.DataParallelContext.from_env(backend='gloo')
    x, cov, xu, glm = T.make_inputs(B_GLOBAL, C, seed=11)
    model = T.make_model(C, xu, glm, dp=ctx, dp_gain=dp_gain)
    b = B_GLOBAL // world
    sl = slice(rank * b, (rank + 1) * b)
    loss = model.train_step(torch.zeros(b, dtype=torch.int64), cov[sl], x[sl])          # noise: the shared seeded generator
    g32 = model.optimizer.groups[torch.float32]
    extra = {}
    if dp_gain == 'local':                                       # the gains this rank drew for its slice (a fresh forward, same noise tape)
        gen = torch.Generator().manual_seed(1234)
        noise = {'eps_w': torch.randn(B_GLOBAL, 1, generator=gen), 'eps_d': torch.randn(B_GLOBAL, 32, generator=gen),
                 'eps_beta': torch.randn(C, B_GLOBAL, generator=gen)}
        with torch.no_grad():
            extra['task_var'] = model.forward_core(cov[sl], x[sl], noise=noise)['task_var'].clone()
    torch.save({'loss': float(loss), 'g': g32['g'].clone(), 'p': g32['p'].clone(), 'eps': model.epsilon.detach().clone(), **extra},
               os.path.join(out_dir, 'rank%d.pt' % rank))
    ctx.shutdown()


@pytest.fixture(autouse=True)
def restore_library():
    prev = _lib._LIB
    yield
    _lib._LIB = prev


def test_two_ranks_equal_one_rank_global_batch(tmp_path):
    ref_loss, ref_g, ref_p, ref_eps = _single_process_reference()
    port = _free_port()
    mp.spawn(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [torch.load(os.path.join(tmp_path, 'rank%d.pt' % r)) for r in range(2)]
    for o in outs:
        np.testing.assert_allclose(o['loss'], ref_loss, rtol=2e-5)
        gn = float(ref_g.norm())
        assert float((o['g'] - ref_g).norm()) <= 2e-4 * gn, (float((o['g'] - ref_g).norm()), gn)     # fp32 reduction order only
        np.testing.assert_allclose(o['eps'].numpy(), ref_eps.numpy(), atol=1e-9)
    assert torch.equal(outs[0]['p'], outs[1]['p'])                  # replicas stay identical
    moved = (ref_p - outs[0]['p']).abs()
    assert float(moved.max()) <= 2.1e-3                             # differences only where a near-zero gradient changed sign


def test_local_gain_mode_draws_each_ranks_slice_from_its_own_covariance(tmp_path):
    """dp_gain='local' (an option; the default is the exact joint draw, 'global'): a rank draws the gains of ITS slice from that slice's own B x B gain
    covariance (its columns of the noise tape; the block-diagonal approximation of the joint draw, cost independent of the number of
    ranks), and the HRF of the neural covariates then runs along the GLOBAL batch index across the slices (ops.HrfAcrossRanks) -- it
    does not restart at a rank boundary.  Batch-norm statistics, the loss normalisation and the gradient sum stay global: replicas
    remain identical."""
    port = _free_port()
    C = 8                                                        # the full covariate set (HRF on task, 6 GP regressors, sex) under data parallelism
    mp.spawn(_rank_main, args=(2, port, str(tmp_path), 'local', C), nprocs=2, join=True)
    outs = [torch.load(os.path.join(tmp_path, 'rank%d.pt' % r)) for r in range(2)]
    assert torch.equal(outs[0]['p'], outs[1]['p']) and torch.equal(outs[0]['g'], outs[1]['g'])
    T.load_emu_library()
    x, cov, xu, glm = T.make_inputs(B_GLOBAL, C, seed=11)
    model = T.make_model(C, xu, glm)                             # same seed: the parameters the ranks started from
    gen = torch.Generator().manual_seed(1234)
    torch.randn(B_GLOBAL, 1, generator=gen); torch.randn(B_GLOBAL, 32, generator=gen)
    eps_beta = torch.randn(C, B_GLOBAL, generator=gen)
    b = B_GLOBAL // 2
    # the ranks evaluated their gains AFTER one optimiser step: bring the single-process model to the same parameters
    model.optimizer.groups[torch.float32]['p'].copy_(outs[0]['p'])
    with torch.no_grad():
        pre = torch.cat([model._gains(cov[r * b:(r + 1) * b][:, :C].float(), eps_beta[:, r * b:(r + 1) * b].contiguous(), hrf_in_kernel=False)[0]
                         for r in range(2)], 1)                  # (C, B_GLOBAL): per-slice draws, not convolved
        want = pre.clone()
        hrf_rows = [i for i, c in enumerate(model.schema) if c.hrf]
        assert hrf_rows == [0]
        for i in hrf_rows:
            want[i] = model.do_hrf_conv(pre[i])                  # ONE causal convolution along all B_GLOBAL volumes (vae_reg_GP.py:283-305)
        restart = torch.cat([model.do_hrf_conv(pre[0, :b]), model.do_hrf_conv(pre[0, b:])])
    assert float((want[0, b:] - restart[b:]).abs().max()) > 1e-3             # the case does tell the two apart
    for r in range(2):
        np.testing.assert_allclose(outs[r]['task_var'].numpy(), want[:, r * b:(r + 1) * b].numpy(), rtol=1e-5, atol=1e-6)


def _hrf_rank(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from vae_gam_amd import dp as dpmod, ops
    ctx = dpmod.DataParallelContext.from_env(backend='gloo')
    g = torch.Generator().manual_seed(5)
    Bg, Ch = 12, 2
    b = Bg // world
    pre_full = torch.randn(Ch, Bg, generator=g); w_full = torch.randn(Ch, Bg, generator=g); Tm = torch.randn(Bg, Bg, generator=g).triu()
    pre = pre_full[:, rank * b:(rank + 1) * b].clone().requires_grad_(True)
    out = ops.HrfAcrossRanks.apply(pre, Tm, ctx, rank * b)
    (out * w_full[:, rank * b:(rank + 1) * b]).sum().backward()
    torch.save({'out': out.detach(), 'g': pre.grad}, os.path.join(out_dir, 'hrf%d.pt' % rank))
    ctx.shutdown()


def test_hrf_across_ranks_forward_and_gradient_equal_the_global_convolution(tmp_path):
    """ops.HrfAcrossRanks on 2 ranks == the (Bg, Bg) Toeplitz product of one process on the concatenated gains, forward and gradient
    (the gradient of a rank's loss with respect to gains ANOTHER rank drew travels through the all-reduce)."""
    port = _free_port()
    mp.spawn(_hrf_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = torch.Generator().manual_seed(5)
    pre = torch.randn(2, 12, generator=g).requires_grad_(True); w = torch.randn(2, 12, generator=g); Tm = torch.randn(12, 12, generator=g).triu()
    out = pre @ Tm
    (out * w).sum().backward()
    for r in range(2):
        o = torch.load(os.path.join(tmp_path, 'hrf%d.pt' % r))
        np.testing.assert_allclose(o['out'].numpy(), out.detach()[:, r * 6:(r + 1) * 6].numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(o['g'].numpy(), pre.grad[:, r * 6:(r + 1) * 6].numpy(), rtol=1e-6, atol=1e-6)


def test_device_resident_data_shards_every_global_batch():
    from vae_gam_amd.DataClass_GP import DeviceResidentData
    N = 20
    vols = torch.arange(N, dtype=torch.float32).view(N, 1, 1, 1).expand(N, 2, 2, 2).contiguous()
    cov = torch.arange(N, dtype=torch.float32).view(N, 1).repeat(1, 3)
    sid = torch.zeros(N, dtype=torch.int64)
    full = [b['covariates'][:, 0].tolist() for b in DeviceResidentData(vols, cov, sid, batch_size=8, shuffle=True, seed=3)]
    parts = [[b['covariates'][:, 0].tolist() for b in DeviceResidentData(vols, cov, sid, batch_size=8, shuffle=True, seed=3, rank=r, world=2)]
             for r in range(2)]
    assert len(full) == 2 and all(len(b) == 8 for b in full)
    for i, batch in enumerate(full):
        assert parts[0][i] + parts[1][i] == batch
