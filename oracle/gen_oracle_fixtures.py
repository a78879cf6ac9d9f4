#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: oracle outputs at sizes that are too slow to recompute inside the GPU tests.

Runs the CPU oracle (fp32 = the reference's arithmetic, and the same algorithm in float64 as a
conditioning yardstick) on the BASELINE config-2 shape (batch 32, 3 covariates, synthetic checker
set) and stores what tests/test_model_gpu.py compares against: loss terms, gains, and for every
parameter the gradient norm, the fp32-vs-fp64 distance and up to 512 sampled entries.
Weights come from the product's seeded initialiser, inputs from vae_gam_amd.synthetic (both are
recipes: only seeds travel).   Usage: python oracle/gen_oracle_fixtures.py
"""
import os
import sys
import time
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
import bridge  # noqa: E402
import vaegam_oracle as O  # noqa: E402
import vae_gam_amd  # noqa: E402,F401
from vae_gam_amd import synthetic  # noqa: E402
from vae_gam_amd.vae_reg_GP import VAE  # noqa: E402


def case_inputs(B=32, C=3, data_seed=3, noise_seed=5, model_seed=1, device='cpu'):
    ds = synthetic.make_dataset(num_subjects=2, vols_per_subject=max(20, B // 2), num_covariates=C, seed=data_seed)
    torch.manual_seed(model_seed)
    model = VAE(num_covariates=C, glm_maps=ds['glm'], xu_ranges=ds['xu_ranges'], device_name=device)
    x = torch.from_numpy(ds['volumes'][:B]); cov = torch.from_numpy(ds['covariates'][:B])
    cfg = bridge.oracle_config(model)
    noise = O.draw_noise(B, cfg, torch.Generator().manual_seed(noise_seed))
    return ds, model, cfg, x, cov, noise


def hires_inputs(B=2, C=12, img=(82, 98, 70), seed=9, noise_seed=6, model_seed=1, device='cpu', n_ind=6, gp_jitter=0.0):
    """BASELINE configs[4] geometry (82x98x70, 12 covariates; no reference counterpart) at a tiny batch.
    n_ind = 64 with gp_jitter > 0 is configs[4] as stated (64 inducing points): the reference's plain inverse of Ku is singular
    there in any precision; the jittered Cholesky form is the build's documented remedy."""
    rng = np.random.Generator(np.random.PCG64(seed))
    V = int(np.prod(img))
    x = torch.from_numpy(np.clip(0.5 + 0.25 * rng.standard_normal((B,) + img), 0, 1).astype(np.float32))
    cont = rng.standard_normal((B, C - 2)); cont[0] = 6.0; cont[1] = -4.0
    cov = np.concatenate([(np.arange(B) % 2)[:, None], cont, (np.arange(B) % 2)[:, None]], 1).astype(np.float32)
    xu = [[float(cont[:, j].min()) - 1e-3, float(cont[:, j].max()) + 1e-3] for j in range(C - 2)]
    glm = rng.uniform(size=(V, C)); glm = glm / glm.max(0, keepdims=True)
    glm = np.concatenate([np.arange(V, dtype=np.float64)[:, None], glm], 1)
    torch.manual_seed(model_seed)
    model = VAE(num_covariates=C, glm_maps=glm, xu_ranges=xu, device_name=device, img_shape=img, num_inducing_pts=n_ind, gp_jitter=gp_jitter)
    cfg = bridge.oracle_config(model)
    noise = O.draw_noise(B, cfg, torch.Generator().manual_seed(noise_seed))
    return glm, model, cfg, x, torch.from_numpy(cov), noise


def sample_idx(name, n, k=512):
    r = np.random.Generator(np.random.PCG64(zlib.crc32(name.encode())))
    return np.sort(r.choice(n, min(k, n), replace=False))


def write_case(name, B, C):
    """fp32 (= the reference's arithmetic) and float64 oracle outputs of one train step at batch B, C covariates."""
    ds, model, cfg, x, cov, noise = case_inputs(B=B, C=C)
    params = bridge.params_from_model(model)
    glm = torch.from_numpy(ds['glm'])
    t = time.time()
    out32, g32 = O.loss_and_grads(params, cfg, x, cov, glm, noise)
    print('%s fp32 oracle %.1fs' % (name, time.time() - t), flush=True)
    t = time.time()
    p64, x64, c64, n64 = O.to_float64(params, x, cov, noise)
    out64, g64 = O.loss_and_grads(p64, cfg, x64, c64, glm, n64)
    print('%s fp64 oracle %.1fs' % (name, time.time() - t), flush=True)
    arr = {'loss32': out32['loss'].detach().numpy(), 'loss64': out64['loss'].detach().numpy(),
           'slp32': out32['sum_log_prob'].detach().numpy(), 'z32': out32['z'].detach().numpy(),
           'kl_z32': out32['kl_z'].detach().numpy(), 'gp_kl32': out32['gp_kl_loss'].detach().numpy(),
           'glm_reg32': out32['glm_reg'].detach().numpy(),
           'slp64': out64['sum_log_prob'].detach().numpy(), 'gp_kl64': out64['gp_kl_loss'].detach().numpy(),
           'glm_reg64': out64['glm_reg'].detach().numpy()}
    for c in cfg.schema:
        arr['task_var32.' + c.name] = out32['task_var'][c.name].detach().numpy()
        arr['task_var64.' + c.name] = out64['task_var'][c.name].detach().numpy()
        if c.gp:
            arr['f_bar64.' + c.name] = out64['f_bar'][c.name].detach().numpy()
            arr['Sigma_diag64.' + c.name] = out64['Sigma'][c.name].detach().diagonal().numpy().copy()
    for k in g32:
        if g32[k] is None:
            continue
        a32 = g32[k].double().flatten().numpy(); a64 = g64[k].flatten().numpy()
        idx = sample_idx(k, a64.size)
        arr['g.%s.norm64' % k] = np.sqrt((a64 * a64).sum())
        arr['g.%s.dist32_64' % k] = np.sqrt(((a32 - a64) ** 2).sum())
        arr['g.%s.idx' % k] = idx
        arr['g.%s.val64' % k] = a64[idx]
        arr['g.%s.val32' % k] = a32[idx]
    out = os.path.join(ROOT, 'tests', 'golden', name + '.npz')
    np.savez_compressed(out, **arr)
    print('wrote', out, os.path.getsize(out))


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    only = sys.argv[1:]
    if not only or 'B32_C3' in only:
        write_case('oracle_B32_C3', 32, 3)
    if not only or 'B64_C8' in only:
        write_case('oracle_B64_C8', 64, 8)          # BASELINE configs[2]: the headline workload of bench.py
    if not only or 'hires_n64' in only:
        # configs[4] as stated: 64 inducing points -- float64 oracle with the jittered Ku (the yardstick; fp32 too, for scale)
        glm, model, cfg, x, cov, noise = hires_inputs(n_ind=64, gp_jitter=1e-4)
        params = bridge.params_from_model(model)
        t = time.time()
        out32, g32 = O.loss_and_grads(params, cfg, x, cov, torch.from_numpy(glm), noise)
        p64, x64, c64, n64 = O.to_float64(params, x, cov, noise)
        out64, g64 = O.loss_and_grads(p64, cfg, x64, c64, torch.from_numpy(glm), n64)
        print('hi-res n=64 fp32 + fp64 oracle %.1fs' % (time.time() - t), flush=True)
        arr = {'loss32': out32['loss'].detach().numpy(), 'loss64': out64['loss'].detach().numpy(),
               'slp64': out64['sum_log_prob'].detach().numpy(), 'z64': out64['z'].detach().numpy(),
               'gp_kl64': out64['gp_kl_loss'].detach().numpy()}
        for c in cfg.schema:
            arr['task_var64.' + c.name] = out64['task_var'][c.name].detach().numpy()
            arr['task_var32.' + c.name] = out32['task_var'][c.name].detach().numpy()
            if c.gp:
                arr['f_bar64.' + c.name] = out64['f_bar'][c.name].detach().numpy()
                arr['Sigma64.' + c.name] = out64['Sigma'][c.name].detach().numpy()
        for k in g64:
            if g64[k] is None or not k.startswith('gp.'):
                continue
            arr['g64.' + k] = g64[k].flatten().numpy()
            arr['g32.' + k] = g32[k].double().flatten().numpy()
        out = os.path.join(ROOT, 'tests', 'golden', 'oracle_hires_B2_C12_n64.npz')
        np.savez_compressed(out, **arr)
        print('wrote', out, os.path.getsize(out))
    if only and 'hires' not in only:
        return
    # ---- hi-res geometry, fp32 oracle only
    glm, model, cfg, x, cov, noise = hires_inputs()
    params = bridge.params_from_model(model)
    t = time.time()
    out32, g32 = O.loss_and_grads(params, cfg, x, cov, torch.from_numpy(glm), noise)
    print('hi-res fp32 oracle %.1fs' % (time.time() - t), flush=True)
    arr = {'loss32': out32['loss'].detach().numpy(), 'slp32': out32['sum_log_prob'].detach().numpy(),
           'z32': out32['z'].detach().numpy(), 'kl_z32': out32['kl_z'].detach().numpy()}
    for k in g32:
        if g32[k] is None:
            continue
        a32 = g32[k].double().flatten().numpy()
        idx = sample_idx(k, a32.size, 256)
        arr['g.%s.norm32' % k] = np.sqrt((a32 * a32).sum()); arr['g.%s.idx' % k] = idx; arr['g.%s.val32' % k] = a32[idx]
    out = os.path.join(ROOT, 'tests', 'golden', 'oracle_hires_B2_C12.npz')
    np.savez_compressed(out, **arr)
    print('wrote', out, os.path.getsize(out))


if __name__ == '__main__':
    main()
