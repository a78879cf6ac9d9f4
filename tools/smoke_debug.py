import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
import vae_gam_amd
from vae_gam_amd import synthetic
from vae_gam_amd.vae_reg_GP import VAE
import bridge, vaegam_oracle as O
B, C = 4, 3
ds = synthetic.make_dataset(num_subjects=1, vols_per_subject=8, num_covariates=C, seed=2)
torch.manual_seed(1)
model = VAE(num_covariates=C, glm_maps=ds['glm'], xu_ranges=ds['xu_ranges'], device_name='cuda')
x = torch.from_numpy(ds['volumes'][:B]); cov = torch.from_numpy(ds['covariates'][:B])
cfg = bridge.oracle_config(model); params = bridge.params_from_model(model)
noise = O.draw_noise(B, cfg, torch.Generator().manual_seed(3))
torch.set_num_threads(16)
out, grads = O.loss_and_grads(params, cfg, x, cov, torch.from_numpy(ds['glm']), noise)
p64, x64, c64, n64 = O.to_float64(params, x, cov, noise)
out64, grads64 = O.loss_and_grads(p64, cfg, x64, c64, torch.from_numpy(ds['glm']), n64)
ids = torch.zeros(B, dtype=torch.int64, device='cuda')
model.optimizer.zero_grad()
loss = model.forward(ids, cov.cuda(), x.cuda(), 'train', noise=bridge.noise_to(noise, 'cuda'))
loss.backward()
print('loss', float(loss), float(out['loss']), float(out64['loss']))
by = bridge.model_param_by_oracle_name(model)
for k, r in grads.items():
    if r is None: continue
    g = by[k].grad.detach().cpu().double().flatten().numpy(); r32 = r.double().flatten().numpy(); r64 = grads64[k].flatten().numpy()
    n = np.sqrt((r64*r64).sum())
    print('%-18s |g64| %.4g  hip-64 %.3g  o32-64 %.3g' % (k, n, np.sqrt(((g-r64)**2).sum())/max(n,1e-30), np.sqrt(((r32-r64)**2).sum())/max(n,1e-30)))
