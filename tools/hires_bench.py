"""Throughput of the proposed 82x98x70 / 12-covariate geometry (BASELINE configs[4] shape, per-GPU share) on one MI355X."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vae_gam_amd  # noqa: F401
from vae_gam_amd import synthetic
from vae_gam_amd.vae_reg_GP import VAE

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_ind = int(sys.argv[2]) if len(sys.argv) > 2 else 64
jit = float(sys.argv[3]) if len(sys.argv) > 3 else (1e-4 if n_ind > 12 else 0.0)      # dense inducing grids need the H2 remedy
ds = synthetic.make_dataset(num_subjects=1, vols_per_subject=max(B, 16), num_covariates=12, img_shape=(82, 98, 70), seed=0)
torch.manual_seed(1)
m = VAE(num_covariates=12, num_inducing_pts=n_ind, glm_maps=ds['glm'], xu_ranges=ds['xu_ranges'], device_name='cuda', img_shape=(82, 98, 70), gp_jitter=jit)
m.use_hip_graph = True
vol = torch.from_numpy(ds['volumes'][:B]).cuda(); cov = torch.from_numpy(ds['covariates'][:B]).cuda(); sid = torch.from_numpy(ds['subjid'][:B]).cuda()
for _ in range(3):
    loss = m.train_step(sid, cov, vol)
torch.cuda.synchronize()
t0 = time.perf_counter(); K = 10
for _ in range(K):
    loss = m.train_step(sid, cov, vol)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print('82x98x70 C=12 n=%d jitter %g batch %d: %.1f ms/step, %.1f volumes/s, loss %.1f, peak mem %.1f GB, graph=%s'
      % (n_ind, jit, B, dt * 1e3, B / dt, float(loss), torch.cuda.max_memory_allocated() / 2**30, bool(m._graphs) and all(v is not False for v in m._graphs.values())))
