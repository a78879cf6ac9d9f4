"""Which part of the train step refuses hipGraph stream capture?  (diagnostic, run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vae_gam_amd
from vae_gam_amd import synthetic
from vae_gam_amd.vae_reg_GP import VAE

ds = synthetic.make_dataset(num_subjects=1, vols_per_subject=40, num_covariates=3, seed=0)
torch.manual_seed(1)
m = VAE(num_covariates=3, glm_maps=ds['glm'], xu_ranges=ds['xu_ranges'], device_name='cuda')
B = 32
x = torch.from_numpy(ds['volumes'][:B]).cuda(); cov = torch.from_numpy(ds['covariates'][:B]).cuda()
ids = torch.zeros(B, dtype=torch.int64, device='cuda')


def probe(name, fn):
    try:
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn(); fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        g.replay(); torch.cuda.synchronize()
        print('CAPTURE OK  ', name, flush=True)
    except Exception as e:
        print('CAPTURE FAIL', name, type(e).__name__, str(e).split('\n')[0], flush=True)
        torch.cuda.synchronize()


noise = m.draw_noise(B, x.device)
probe('randn', lambda: m.draw_noise(B, x.device))
probe('inv_ex f64', lambda: torch.linalg.inv_ex(torch.eye(6, device='cuda', dtype=torch.float64).expand(2, 6, 6) * 2, check_errors=False))
from vae_gam_amd import ops
probe('vg cholesky f64', lambda: ops.cholesky((torch.eye(32, device='cuda', dtype=torch.float64).expand(3, 32, 32) * 2).contiguous()))
probe('solve_triangular f64', lambda: torch.linalg.solve_triangular(torch.eye(32, device='cuda', dtype=torch.float64).expand(3, 32, 32).contiguous(), torch.ones(3, 32, 32, device='cuda', dtype=torch.float64), upper=False))
probe('gains', lambda: m._gains(cov, noise['eps_beta']))
probe('encode', lambda: m.encode(x))
probe('forward_core', lambda: m.forward_core(cov, x, noise))
def fb():
    m.optimizer.zero_grad(); l = m.forward(ids, cov, x, 'train', noise=noise); l.backward()
probe('fwd+bwd', fb)
probe('adam', lambda: m.optimizer.apply_update())
W = torch.randn(17, 40, device='cuda'); Wg = torch.zeros_like(W); bgr = torch.zeros(17, device='cuda')
gy = torch.randn(96, 17, device='cuda'); xx = torch.randn(96, 40, device='cuda'); one = torch.ones(96, device='cuda')
probe('addmm_', lambda: Wg.addmm_(gy.t(), xx))
probe('addmv_', lambda: bgr.addmv_(gy.t(), one))
probe('threshold_backward', lambda: torch.ops.aten.threshold_backward(gy, gy, 0.0))
mu = torch.randn(32, 32, device='cuda'); ew = torch.randn(32, 1, device='cuda')
probe('latent', lambda: ops.LatentSample.apply(mu, mu, mu, ew, mu, 4))
kl = torch.randn(32, device='cuda'); dist = torch.randn(3, 32, device='cuda'); gp_ = torch.randn(1, device='cuda')
probe('loss', lambda: ops.ElboLoss.apply(kl, kl, dist, gp_, (1.0, 2.0, 3.0, 4.0)))
zc = torch.randn(128, 36, device='cuda')
probe('decode', lambda: m._decode_logits(zc, 32))
def core_nograd():
    with torch.no_grad():
        m.forward_core(cov, x, noise)
probe('forward_core nograd', core_nograd)
