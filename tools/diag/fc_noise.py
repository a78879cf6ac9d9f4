"""Sensitivity of the B=4 golden step's gradients to the SUMMATION ORDER inside the fully connected products (MI355X): the same step with
the default split-K plan and with split-K off; prints, per parameter, the relative change of the gradient and of the sampled entries."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
import test_model_gpu as T
from vae_gam_amd import ops
gd = os.path.join(ROOT, 'tests', 'golden')
out = {}
for mode in ('default', 'nosplit'):
    if mode == 'nosplit':
        ops._fc_split = lambda M, N, K: 1
    g, meta, model, x, cov, noise, noise2, glm = T.build_from_golden(gd, sys.argv[1] if len(sys.argv) > 1 else 'ref_B4_C8')
    ids = torch.zeros(x.shape[0], dtype=torch.int64, device='cuda')
    model.optimizer.zero_grad()
    loss = model.forward(ids, cov, x, 'train', train_mode=True, noise=noise)
    loss.backward(); torch.cuda.synchronize()
    out[mode] = {n: p.grad.detach().double().cpu().numpy().ravel().copy() for n, p in model.named_parameters() if p.grad is not None}
for n in out['default']:
    a, b = out['default'][n], out['nosplit'][n]
    d = np.abs(a - b)
    if np.linalg.norm(a) > 0:
        print('%-16s |g| %10.3e  rel norm diff %8.1e   max entry diff / rms %8.1e' % (n, np.linalg.norm(a), np.linalg.norm(a - b) / np.linalg.norm(a), d.max() / (np.linalg.norm(a) / np.sqrt(a.size))))
import bridge
byname = bridge.model_param_by_oracle_name(model)
names = {id(p): n for n, p in model.named_parameters()}
print('--- against the reference golden (sampled entries): max |diff| / (rtol*|ref| + atol) per parameter, default plan')
for k, p in byname.items():
    if ('grad.%s.none' % k) in g or k.endswith(('.logkvar', '.log_ls')):
        continue
    gf = out['default'][names[id(p)]]
    ref_norm = float(g['grad.%s.norm' % k]); idx = g['grad.%s.idx' % k]; val = g['grad.%s.val' % k]
    tol = 2e-3 * np.abs(val) + 1e-6 + 1e-3 * ref_norm / np.sqrt(gf.size)
    r = np.abs(gf[idx] - val) / tol
    if r.max() > 0.3:
        print('%-22s worst %.2f of tolerance; rel norm err %.1e' % (k, r.max(), abs(np.linalg.norm(gf) - ref_norm) / ref_norm))
