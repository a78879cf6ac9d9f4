"""Gain block (vg_gp_gain_fwd / _bwd) alone at data-parallel global batch sizes:  python tools/diag/gain_bench.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import vae_gam_amd
from vae_gam_amd import synthetic
from vae_gam_amd.vae_reg_GP import VAE
Bs = [int(v) for v in sys.argv[1:]] or [64, 128, 256, 512]
ds = synthetic.make_dataset(num_subjects=8, vols_per_subject=98, num_covariates=8, seed=0)
torch.manual_seed(1)
m = VAE(num_covariates=8, glm_maps=ds['glm'], xu_ranges=ds['xu_ranges'], device_name='cuda')
m.overlap_gains = False
for B in Bs:
    cov = torch.from_numpy(ds['covariates'][:B]).cuda()
    eps = torch.randn(8, B, device='cuda')
    def fwd():
        return m._gains(cov, eps)
    def both():
        m.optimizer.zero_grad()
        tv, kl = fwd()[:2]
        (tv.sum() + kl.sum()).backward()
    for fn, nm in ((fwd, 'fwd'), (both, 'fwd+bwd')):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        print('B %4d  %-8s %8.3f ms' % (B, nm, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
