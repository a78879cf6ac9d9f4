"""Train a few hundred steps on the synthetic checker set (hipGraph path) and print the loss trajectory (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vae_gam_amd  # noqa: F401
from vae_gam_amd import synthetic
from vae_gam_amd.vae_reg_GP import VAE

ds = synthetic.make_dataset(num_subjects=2, vols_per_subject=98, num_covariates=3, seed=0)
torch.manual_seed(1)
m = VAE(num_covariates=3, glm_maps=ds['glm'], xu_ranges=ds['xu_ranges'], device_name='cuda')
m.use_hip_graph = True
B = 32
vol = torch.from_numpy(ds['volumes']).cuda(); cov = torch.from_numpy(ds['covariates']).cuda(); sid = torch.from_numpy(ds['subjid']).cuda()
g = torch.Generator().manual_seed(0)
losses = []
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    idx = torch.randperm(vol.shape[0], generator=g)[:B].cuda()
    losses.append(m.train_step(sid[idx], cov[idx], vol[idx]).clone())
torch.cuda.synchronize()
l = torch.cat(losses).cpu()
assert torch.isfinite(l).all(), 'non-finite loss'
print('loss: first %.1f  step50 %.1f  step150 %.1f  last %.1f  (min %.1f)' % (l[0], l[min(50, len(l) - 1)], l[min(150, len(l) - 1)], l[-1], l.min()))
assert l[-20:].mean() < l[:20].mean(), 'loss did not go down'
print('OK')
