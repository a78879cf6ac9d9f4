#!/usr/bin/env python3
"""Golden-vector generator -- runs the REFERENCE itself (in the build container only).

TEST INFRASTRUCTURE.  Imports dannyfa/VAE-GAM's own `vae_reg_GP.VAE` / `gp.GP` from
/root/reference (never copied, never edited), through harness-side stubs for the modules
that are absent from this image and do not touch the arithmetic (tensorboard SummaryWriter,
nibabel, umap, torchvision), plus one shim: `gp._striped_matrix` hard-codes `.cuda()`
(gp.py:115), which is replaced by an equivalent CPU builder.

Inputs are seeded recipes; the noise the reference draws through
`torch.distributions.utils._standard_normal` is injected so the same draws can be replayed
by the oracle restatement and by the HIP path.  Outputs (loss terms, latents, GP posteriors,
gains, map statistics, gradients, post-Adam parameters, checkpoint key listing) are written
as small .npz/.json fixtures under tests/golden/.

The reference does not exist on the GPU box; only the fixtures travel.

Usage:  python oracle/gen_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import json
import os
import sys
import tempfile
import types
import zlib

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import vaegam_oracle as O  # noqa: E402


# ----------------------------------------------------------------------------- stubs
def install_stubs():
    class _NoWriter:
        def __init__(self, *a, **k): pass
        def __getattr__(self, name):
            return lambda *a, **k: None
    tb = types.ModuleType('torch.utils.tensorboard'); tb.SummaryWriter = _NoWriter
    sys.modules['torch.utils.tensorboard'] = tb
    sys.modules['tensorboard'] = types.ModuleType('tensorboard')
    sys.modules['nibabel'] = types.ModuleType('nibabel')
    um = types.ModuleType('umap'); um.UMAP = object
    sys.modules['umap'] = um
    tv = types.ModuleType('torchvision'); tvd = types.ModuleType('torchvision.datasets')
    tvt = types.ModuleType('torchvision.transforms'); tv.datasets, tv.transforms = tvd, tvt
    sys.modules.update({'torchvision': tv, 'torchvision.datasets': tvd, 'torchvision.transforms': tvt})


def import_reference(ref_dir):
    install_stubs()
    sys.path.insert(0, ref_dir)
    import gp as ref_gp
    import vae_reg_GP as ref_vae

    def striped_cpu(n):      # same values as gp._striped_matrix (gp.py:113-119) without .cuda()
        idx = torch.arange(n, dtype=torch.float32)
        return (idx.unsqueeze(0) - idx.unsqueeze(1)).abs()
    ref_gp._striped_matrix = striped_cpu
    return ref_vae, ref_gp


# ----------------------------------------------------------------------------- inputs
def make_case_inputs(seed, B, C, img=(41, 49, 35)):
    """Seeded recipe (numpy PCG64) for one minibatch + the CSVs the constructor reads."""
    rng = np.random.Generator(np.random.PCG64(seed))
    V = int(np.prod(img))
    T = max(2 * B, 16)
    # wide-range continuous covariates so that Ku stays well conditioned
    cont = rng.normal(size=(T, 6))
    cont[0] = 6.0 + 0.1 * rng.normal(size=6)
    cont[1] = -4.0 + 0.1 * rng.normal(size=6)
    task = (np.arange(T) // 3 % 2 == 0).astype(np.float64)
    sex = (np.arange(T) >= T // 2).astype(np.float64)
    df = pd.DataFrame({'subjid': ['s%d' % (i >= T // 2) for i in range(T)], 'volume #': np.arange(T) % (T // 2),
                       'nii_path': ['none'] * T, 'task': task, 'x': cont[:, 0], 'y': cont[:, 1], 'z': cont[:, 2],
                       'rot_x': cont[:, 3], 'rot_y': cont[:, 4], 'rot_z': cont[:, 5], 'sex': sex})
    full_cov = np.stack([task, *cont.T, sex], 1)                       # reference layout (DataClass_GP.py:66-67)
    rows = rng.permutation(T)[:B]
    rows[:2] = [0, 1]                                                  # keep the range-setting outliers in the batch
    covariates = full_cov[rows][:, :C].astype(np.float32)
    x = np.clip(0.5 + 0.25 * rng.normal(size=(B,) + tuple(img)), 0, 1).astype(np.float32)
    glm = rng.uniform(size=(V, 8))
    glm = glm / glm.max(0, keepdims=True)
    glm_df = pd.DataFrame(glm, columns=REF_GLM_COLS)
    return dict(df=df, covariates=covariates, x=x, glm_df=glm_df)


REF_GLM_COLS = ['task', 'x', 'y', 'z', 'rot_x', 'rot_y', 'rot_z', 'sex']


class NoiseTape:
    """Replaces torch.distributions' _standard_normal; records every draw."""
    def __init__(self, seed):
        self.gen = torch.Generator().manual_seed(seed)
        self.draws = []

    def __call__(self, shape, dtype, device):
        t = torch.randn(tuple(shape), generator=self.gen, dtype=dtype)
        self.draws.append(t.clone())
        return t


def patch_noise(tape):
    import torch.distributions.lowrank_multivariate_normal as lr
    import torch.distributions.multivariate_normal as mv
    lr._standard_normal = tape
    mv._standard_normal = tape


def map_stats(m, vox):
    m = m.astype(np.float64)
    return np.concatenate([[m.sum(), (m * m).sum()], m[:, vox].ravel()])


# ----------------------------------------------------------------------------- one case
def run_case(ref_vae, name, seed, B, C, out_dir, neural=True):
    inp = make_case_inputs(seed, B, C)
    tmp = tempfile.mkdtemp(prefix='vg_golden_')
    train_csv, test_csv, glm_csv = [os.path.join(tmp, f) for f in ('train.csv', 'test.csv', 'glm.csv')]
    inp['df'].to_csv(train_csv); inp['df'].iloc[:4].to_csv(test_csv); inp['glm_df'].to_csv(glm_csv)

    torch.manual_seed(1)                                               # CLI default (multsubj_reg_run_GP.py:31,57)
    model = ref_vae.VAE(num_covariates=C, glm_maps=glm_csv, save_dir=tmp, csv_files=[train_csv, test_csv],
                        neural_covariates=neural)
    assert model.device.type == 'cpu'

    # ---- oracle parameters drawn with the same seed must equal the reference's own init
    cfg = O.OracleConfig(num_covariates=C, neural_covariates=neural)
    xu_ranges = sys.modules['utils'].get_xu_ranges([train_csv, test_csv])
    torch.manual_seed(1)
    p = O.init_params(cfg, xu_ranges)
    ref_named = dict(model.named_parameters())
    name_map = {}                                                      # oracle name -> reference tensor
    name_map['epsilon'] = model.epsilon
    for gname, d in model.gp_params.items():
        for k, v in d.items():
            name_map['gp.%s.%s' % (gname, k)] = v
    for lname, layer in model._get_layers().items():
        name_map[lname + '.weight'], name_map[lname + '.bias'] = layer.weight, layer.bias
    init_max = 0.0
    for k, v in p.items():
        init_max = max(init_max, float((v.double() - name_map[k].detach().double()).abs().max()))
    assert init_max == 0.0, 'oracle init differs from reference init: %g' % init_max

    # ---- reference forward / backward / Adam with recorded noise
    tape = NoiseTape(seed + 1000)
    patch_noise(tape)
    x = torch.from_numpy(inp['x']); cov = torch.from_numpy(inp['covariates'])
    ids = torch.zeros(B, dtype=torch.int64)
    model.train()
    # the gains the reference forms inside forward (vae_reg_GP.py:368-380) are not returned: record them where they pass through
    # torch -- the MultivariateNormal it constructs (loc = beta_mean, rsample = the gain before the HRF) and the first operand of
    # the einsum that scales the effect map (= the gain that is actually applied, after the HRF where it applies)
    rec = {'beta_mean': [], 'pre_hrf': [], 'applied': []}
    MVN = ref_vae.MultivariateNormal

    class RecordingMVN(MVN):
        def __init__(self, loc, covariance_matrix=None, **kw):
            super().__init__(loc, covariance_matrix, **kw)
            rec['beta_mean'].append(loc.detach().clone())

        def rsample(self, sample_shape=torch.Size()):
            t = super().rsample(sample_shape)
            rec['pre_hrf'].append(t.detach().clone())
            return t
    real_einsum = torch.einsum

    def recording_einsum(eq, *ops):
        if eq == 'b,bx->bx':
            rec['applied'].append(ops[0].detach().clone())
        return real_einsum(eq, *ops)
    ref_vae.MultivariateNormal = RecordingMVN
    torch.einsum = recording_einsum
    try:
        loss, z, imgs = model.forward(ids, cov, x, 'train', return_latent_rec=True, train_mode=False)
    finally:
        ref_vae.MultivariateNormal = MVN
        torch.einsum = real_einsum
    assert len(rec['beta_mean']) == len(rec['pre_hrf']) == len(rec['applied']) == C
    draws = tape.draws
    assert len(draws) == 2 + C and draws[0].shape == (B, 1) and draws[1].shape == (B, 32)
    noise = {'eps_w': draws[0], 'eps_d': draws[1], 'eps_beta': torch.stack(draws[2:2 + C])}
    model.optimizer.zero_grad()
    loss.backward()
    grads = {k: (None if v.grad is None else v.grad.detach().clone()) for k, v in name_map.items()
             if isinstance(v, torch.nn.Parameter)}
    mu, u, d = [t.detach() for t in model.encode(x)]

    # float64 evaluation of the same step by the oracle restatement (pinned by tests/test_oracle_golden.py): the yardstick for
    # quantities where the reference's own fp32 arithmetic is conditioning-limited (GP hyper-parameter gradients)
    glm_full = np.concatenate([np.arange(cfg.V)[:, None].astype(np.float64), inp['glm_df'].to_numpy()], 1)
    p64, x64, c64, n64 = O.to_float64(p, x, cov, noise)
    out64, g64 = O.loss_and_grads(p64, cfg, x64, c64, torch.from_numpy(glm_full), n64)

    # GP posteriors straight from the reference's gp.GP (gp.py:67-110)
    ref_gp = sys.modules['gp']
    gp_out = {}
    for i, cname in enumerate(list(model.gp_params.keys())[:C], start=1):
        if 1 < i < 8:
            gpp = model.gp_params[cname]
            kvar = gpp['logkvar'].exp() + 0.1
            ls = model.max_ls * torch.sigmoid(gpp['log_ls'].exp() + 0.5)
            g = ref_gp.GP(gpp['xu'], kvar, ls, gpp['qu_m'], gpp['qu_S'])
            fb, Sg = g.evaluate_posterior(cov[:, i - 1])
            gp_out[cname] = (fb.detach().numpy(), Sg.detach().numpy(),
                             g.compute_GP_kl(6, i, cov[:, i - 1], tmp).detach().numpy())

    # checkpoint key listing (vae_reg_GP.py:452-471)
    model.save_state('ckpt_probe.tar')
    ck = torch.load(os.path.join(tmp, 'ckpt_probe.tar'), weights_only=False)
    ck_keys = {k: (sorted(v.keys()) if isinstance(v, dict) else type(v).__name__) for k, v in ck.items()}
    ck_keys['gp_params'] = {k: sorted(v.keys()) for k, v in ck['gp_params'].items()}
    ck_keys['optimizer_state'] = {'param_groups_keys': sorted(ck['optimizer_state']['param_groups'][0].keys()),
                                  'n_params': len(ck['optimizer_state']['param_groups'][0]['params'])}
    param_order = [k for k, _ in model.named_parameters()]

    model.optimizer.step()
    post = {k: v.detach().clone() for k, v in name_map.items() if isinstance(v, torch.nn.Parameter)}
    # second forward after the step (exercises the updated GP / epsilon parameters)
    tape2 = NoiseTape(seed + 2000); patch_noise(tape2)
    with torch.no_grad():
        loss2 = model.forward(ids, cov, x, 'train', train_mode=False)

    rng = np.random.Generator(np.random.PCG64(7))
    vox = np.sort(rng.choice(inp['x'][0].size, 64, replace=False))
    arrays = {
        'seed': np.array(seed), 'B': np.array(B), 'C': np.array(C), 'neural': np.array(int(neural)),
        'x': None, 'covariates': inp['covariates'], 'vox': vox,
        'eps_w': noise['eps_w'].numpy(), 'eps_d': noise['eps_d'].numpy(), 'eps_beta': noise['eps_beta'].numpy(),
        'eps2_w': tape2.draws[0].numpy(), 'eps2_d': tape2.draws[1].numpy(),
        'eps2_beta': torch.stack(tape2.draws[2:2 + C]).numpy(),
        'loss': loss.detach().numpy(), 'loss2': loss2.detach().numpy(), 'z': z,
        'mu': mu.numpy(), 'u': u.numpy(), 'd': d.numpy(),
    }
    del arrays['x']                                                    # x is a recipe (seed), not stored
    for k in imgs:
        if isinstance(imgs[k], np.ndarray):            # unused covariate slots stay {} (vae_reg_GP.py:308)
            arrays['map.' + k] = map_stats(imgs[k], vox)
    arrays['beta_mean'] = torch.stack(rec['beta_mean']).numpy()            # (C, B)
    arrays['task_var_pre_hrf'] = torch.stack(rec['pre_hrf']).numpy()       # (C, B) MultivariateNormal.rsample(), :369
    arrays['task_var'] = torch.stack(rec['applied']).numpy()               # (C, B) the gain applied to the effect map, :380
    arrays['loss64'] = out64['loss'].detach().numpy()
    for k, g_ in g64.items():
        if g_ is not None and k.startswith('gp.'):
            arrays['grad64.%s' % k] = g_.detach().double().flatten().numpy()
    for cname, (fb, Sg, klv) in gp_out.items():
        arrays['gp.%s.f_bar' % cname], arrays['gp.%s.Sigma' % cname], arrays['gp.%s.kl' % cname] = fb, Sg, klv
    gidx = {}
    for k, g in grads.items():
        if g is None:
            arrays['grad.%s.none' % k] = np.array(1)
            continue
        gf = g.double().flatten().numpy()
        r = np.random.Generator(np.random.PCG64(zlib.crc32(k.encode())))
        idx = np.sort(r.choice(gf.size, min(16, gf.size), replace=False))
        arrays['grad.%s.norm' % k] = np.array(np.sqrt((gf * gf).sum()))
        arrays['grad.%s.idx' % k] = idx
        arrays['grad.%s.val' % k] = gf[idx]
        pf = post[k].double().flatten().numpy()
        arrays['post.%s.sum' % k] = np.array(pf.sum())
        arrays['post.%s.val' % k] = pf[idx]
    np.savez_compressed(os.path.join(out_dir, name + '.npz'), **arrays)
    meta = {'case': name, 'seed': seed, 'B': B, 'C': C, 'neural_covariates': neural, 'model_seed': 1,
            'torch': torch.__version__, 'checkpoint_keys': ck_keys, 'param_order': param_order,
            'loss': float(loss), 'loss2': float(loss2),
            'init_matches_oracle_init_max_abs_diff': init_max}
    with open(os.path.join(out_dir, name + '.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print('%s: loss=%.6f loss2=%.6f  (%d arrays)' % (name, float(loss), float(loss2), len(arrays)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(HERE), 'tests', 'golden'))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.set_num_threads(8)
    ref_vae, _ = import_reference(a.ref)
    run_case(ref_vae, 'ref_B4_C3', seed=11, B=4, C=3, out_dir=a.out)
    run_case(ref_vae, 'ref_B4_C8', seed=12, B=4, C=8, out_dir=a.out)
    run_case(ref_vae, 'ref_B6_C8_nohrf', seed=13, B=6, C=8, out_dir=a.out, neural=False)


if __name__ == '__main__':
    main()
