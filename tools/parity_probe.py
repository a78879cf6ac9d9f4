"""Diagnostic: distances of the HIP path from the reference goldens / the float64 oracle for the quantities the golden test bounds."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
import vae_gam_amd  # noqa
import bridge
from test_model_gpu import build_from_golden
gd = os.path.join(ROOT, 'tests', 'golden')
for name in ['ref_B4_C3', 'ref_B4_C8', 'ref_B6_C8_nohrf']:
    g, meta, model, x, cov, noise, noise2, glm = build_from_golden(gd, name)
    model.optimizer.zero_grad()
    res = model.forward_core(cov, x, noise)
    res['loss'].backward()
    print('==', name)
    for k in ('mu', 'u', 'd'):
        print(' %s max abs err %.2e' % (k, np.abs(res[k].detach().cpu().numpy() - g[k]).max()))
    print(' task_var err', np.abs(res['task_var'].detach().cpu().numpy() - g['task_var']).max(), 'scale', np.abs(g['task_var']).max())
    print(' beta_mean err', np.abs(res['beta_mean'].detach().cpu().numpy() - g['beta_mean']).max())
    if res['gp_post'] is not None:
        gn, fb, Sg = res['gp_post']
        for i, n in enumerate(gn):
            print('  gp %s f_bar %.2e Sigma %.2e' % (n, np.abs(fb[i].detach().cpu().numpy() - g['gp.%s.f_bar' % n]).max(), np.abs(Sg[i].detach().cpu().numpy() - g['gp.%s.Sigma' % n]).max()))
    byname = bridge.model_param_by_oracle_name(model)
    for k, p in byname.items():
        if k.startswith('gp.') and ('grad64.' + k) in g and p.grad is not None:
            a = p.grad.detach().double().cpu().flatten().numpy(); r64 = g['grad64.' + k]
            ref = g.get('grad.%s.val' % k)
            if k.endswith(('.logkvar', '.log_ls', '.sa', '.logstd')):
                print('  %-18s hip %+.5f f64 %+.5f ref32 %s' % (k, a[0], r64[0], ref))
            else:
                print('  %-18s rel err vs f64 %.2e (norm %.3g)' % (k, np.linalg.norm(a - r64) / max(np.linalg.norm(r64), 1e-30), np.linalg.norm(r64)))
    # maps / z / gradient norms (tolerances of SURVEY 8c)
    MAP_KEYS = ['base', 'task', 'x_mot', 'y_mot', 'z_mot', 'pitch_mot', 'roll_mot', 'yaw_mot', 'sex']
    C = model.num_covariates; B = x.shape[0]
    ids = torch.zeros(B, dtype=torch.int64, device='cuda')
    loss, z, imgs = model.forward(ids, cov, x, 'train', return_latent_rec=True, train_mode=False, noise=noise)
    print(' loss rel', abs(float(loss) - float(g['loss'])) / float(g['loss']), ' z abs', np.abs(z - g['z']).max())
    worst = 0
    for key in MAP_KEYS[:C + 1] + ['full_rec']:
        m = imgs[key].astype(np.float64)
        worst = max(worst, np.abs(m[:, g['vox']].ravel() - g['map.' + key][2:]).max())
    print(' maps max abs err at the 64 voxels', worst)
    wn, wk = 0, None; we = 0; wek = None
    for k, p in byname.items():
        if ('grad.%s.none' % k) in g or k.endswith(('.logkvar', '.log_ls')):
            continue
        gf = p.grad.detach().double().flatten().cpu().numpy(); rn = float(g['grad.%s.norm' % k])
        e = abs(np.sqrt((gf * gf).sum()) - rn) / max(rn, 1e-30)
        if e > wn: wn, wk = e, k
        e2 = np.abs(gf[g['grad.%s.idx' % k]] - g['grad.%s.val' % k]).max() / max(rn / np.sqrt(gf.size), 1e-30)
        if e2 > we: we, wek = e2, k
    print(' worst grad-norm rel err %.2e (%s); worst sampled-entry err / rms entry %.2e (%s)' % (wn, wk, we, wek))
