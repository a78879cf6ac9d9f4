"""Run a few train steps of other BASELINE configs on the GPU (does it run, how fast, finite loss?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vae_gam_amd
from vae_gam_amd import synthetic
from vae_gam_amd.vae_reg_GP import VAE

def run(name, C, B, img, n_ind=6, steps=5):
    V = int(np.prod(img))
    rng = np.random.Generator(np.random.PCG64(0))
    if img == (41, 49, 35):
        ds = synthetic.make_dataset(num_subjects=1, vols_per_subject=max(B, 98), num_covariates=C, seed=0)
        x = torch.from_numpy(ds['volumes'][:B]).cuda(); cov = torch.from_numpy(ds['covariates'][:B]).cuda(); glm = ds['glm']; xu = ds['xu_ranges']
    else:
        x = torch.rand((B,) + img, device='cuda'); cont = rng.standard_normal((B, C - 2)); cont[0] = 6; cont[1] = -4
        cov = torch.from_numpy(np.concatenate([(np.arange(B) % 2)[:, None], cont, (np.arange(B) % 2)[:, None]], 1).astype(np.float32)).cuda()
        xu = [[float(cont[:, j].min()) - 1e-3, float(cont[:, j].max()) + 1e-3] for j in range(C - 2)]
        glm = np.concatenate([np.arange(V, dtype=np.float64)[:, None], rng.uniform(size=(V, C))], 1)
    torch.manual_seed(1)
    m = VAE(num_covariates=C, glm_maps=glm, xu_ranges=xu, device_name='cuda', img_shape=img, num_inducing_pts=n_ind)
    ids = torch.zeros(B, dtype=torch.int64, device='cuda')
    l0 = float(m.train_step(ids, cov, x)); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        l = m.train_step(ids, cov, x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps
    print('%-28s C=%d B=%d img=%s: loss %.1f -> %.1f, %.1f ms/step eager = %.0f vol/s, peak mem %.1f GB' %
          (name, C, B, img, l0, float(l), dt * 1e3, B / dt, torch.cuda.max_memory_allocated() / 1e9), flush=True)

if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
    if which == 'cfg3':
        run('configs[2] full model', 8, 64, (41, 49, 35))
    elif which == 'hires':
        run('configs[4]-like hi-res', 12, 4, (82, 98, 70), n_ind=6, steps=2)
