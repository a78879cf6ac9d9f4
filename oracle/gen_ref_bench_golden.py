#!/usr/bin/env python3
"""Golden vectors AT THE BENCH SHAPES from the REFERENCE itself (build container only) -- TEST INFRASTRUCTURE.

BASELINE configs[1] (batch 32, 3 covariates) and configs[2] (batch 64, 8 covariates: bench.py's headline workload) on the
synthetic checker set: the reference's own `vae_reg_GP.VAE` (imported from /root/reference through the stubs of
gen_golden.py, never copied) runs ONE train step -- forward with the recorded noise, backward -- from the same weights,
inputs and noise that tests/test_model_gpu.py hands to the HIP path (recipes: vae_gam_amd.synthetic seed 3, model seed 1,
noise seed 5, as gen_oracle_fixtures.case_inputs).  These are the shapes where the reference's batch-dependent code paths
differ from the small goldens: `torch.cdist` takes its matmul form above 25 rows (vae_reg_GP.py:388), the HRF Toeplitz
product runs along a 32/64-long batch axis (:283-305), the gain covariance is a 64 x 64 Cholesky (:368).

Stored: loss, glm_reg (the sum of the cdist calls), z, the gains (MultivariateNormal loc / rsample, the einsum operand that
scales each effect map), per-map statistics, and for every parameter the gradient norm + the entries at the indices the oracle
fixture samples (gen_oracle_fixtures.sample_idx), so that the float64 yardstick of oracle_B*_C*.npz lines up entry by entry.

Usage:  python oracle/gen_ref_bench_golden.py [--ref /root/reference]      (about 2 minutes of CPU)
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
import gen_golden as G  # noqa: E402
import gen_oracle_fixtures as F  # noqa: E402
import bridge  # noqa: E402


class ReplayTape:
    """Replaces torch.distributions' _standard_normal: hands out the recorded draws in the reference's order."""
    def __init__(self, draws):
        self.draws = list(draws)
        self.k = 0

    def __call__(self, shape, dtype, device):
        t = self.draws[self.k]; self.k += 1
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
        return t.to(dtype).clone()


def run_case(ref_vae, name, B, C, out_dir):
    ds, ours, cfg, x, cov, noise = F.case_inputs(B=B, C=C)              # product model on the CPU: seeded weights only, no kernels run
    params = bridge.params_from_model(ours)
    tmp = tempfile.mkdtemp(prefix='vg_refbench_')
    import pandas as pd
    from vae_gam_amd import synthetic
    csv, glm_csv = synthetic.write_csvs(ds, tmp)
    torch.manual_seed(1)
    model = ref_vae.VAE(num_covariates=C, glm_maps=glm_csv, save_dir=tmp, csv_files=[csv, csv])
    # identical weights: copy OUR seeded parameters into the reference's tensors (they are equal already where the draw order
    # coincides; the copy makes the fixture independent of that), and its inducing grids / GLM maps from the same recipe
    name_map = {'epsilon': model.epsilon}
    for gname, d in model.gp_params.items():
        for k, v in d.items():
            name_map['gp.%s.%s' % (gname, k)] = v
    for lname, layer in model._get_layers().items():
        name_map[lname + '.weight'], name_map[lname + '.bias'] = layer.weight, layer.bias
    init_diff = 0.0
    with torch.no_grad():
        for k, v in params.items():
            tgt = name_map[k]
            init_diff = max(init_diff, float((tgt.detach().double() - v.double()).abs().max()))
            tgt.copy_(v.to(tgt.dtype))
    model.glm_maps = torch.from_numpy(np.ascontiguousarray(ds['glm']))
    print('%s: reference init vs product init max abs diff %.3g (before the copy)' % (name, init_diff), flush=True)

    tape = ReplayTape([noise['eps_w'], noise['eps_d']] + [noise['eps_beta'][i] for i in range(C)])
    G.patch_noise(tape)
    rec = {'beta_mean': [], 'pre_hrf': [], 'applied': [], 'cdist': []}
    MVN = ref_vae.MultivariateNormal

    class RecordingMVN(MVN):
        def __init__(self, loc, covariance_matrix=None, **kw):
            super().__init__(loc, covariance_matrix, **kw)
            rec['beta_mean'].append(loc.detach().clone())

        def rsample(self, sample_shape=torch.Size()):
            t = super().rsample(sample_shape)
            rec['pre_hrf'].append(t.detach().clone())
            return t
    real_einsum, real_cdist = torch.einsum, torch.cdist

    def recording_einsum(eq, *ops):
        if eq == 'b,bx->bx':
            rec['applied'].append(ops[0].detach().clone())
        return real_einsum(eq, *ops)

    def recording_cdist(a, b, p=2.0, **kw):
        out = real_cdist(a, b, p=p, **kw)
        rec['cdist'].append(float(out.detach().double().sum()))
        return out
    ids = torch.zeros(B, dtype=torch.int64)
    model.train()
    ref_vae.MultivariateNormal = RecordingMVN
    torch.einsum, torch.cdist = recording_einsum, recording_cdist
    t0 = time.time()
    try:
        loss, z, imgs = model.forward(ids, cov, x, 'train', return_latent_rec=True, train_mode=False)
    finally:
        ref_vae.MultivariateNormal = MVN
        torch.einsum, torch.cdist = real_einsum, real_cdist
    assert tape.k == 2 + C and len(rec['applied']) == C and len(rec['cdist']) == C
    model.optimizer.zero_grad()
    loss.backward()
    print('%s: reference forward + backward %.1f s, loss %.6f' % (name, time.time() - t0, float(loss)), flush=True)

    rng = np.random.Generator(np.random.PCG64(7))
    vox = np.sort(rng.choice(int(np.prod(x.shape[1:])), 64, replace=False))
    arr = {'B': np.array(B), 'C': np.array(C), 'vox': vox, 'loss': loss.detach().numpy(), 'z': z,
           'glm_reg': np.array(sum(rec['cdist'])), 'glm_reg_terms': np.array(rec['cdist']),
           'beta_mean': torch.stack(rec['beta_mean']).numpy(), 'task_var_pre_hrf': torch.stack(rec['pre_hrf']).numpy(),
           'task_var': torch.stack(rec['applied']).numpy(), 'init_diff': np.array(init_diff)}
    for k in imgs:
        if isinstance(imgs[k], np.ndarray):
            arr['map.' + k] = G.map_stats(imgs[k], vox)
    for k, v in name_map.items():
        if not isinstance(v, torch.nn.Parameter):
            continue
        if v.grad is None:
            arr['grad.%s.none' % k] = np.array(1)
            continue
        gf = v.grad.detach().double().flatten().numpy()
        idx = F.sample_idx(k, gf.size)
        arr['grad.%s.norm' % k] = np.array(np.sqrt((gf * gf).sum()))
        arr['grad.%s.idx' % k] = idx
        arr['grad.%s.val' % k] = gf[idx]
    out = os.path.join(out_dir, name + '.npz')
    np.savez_compressed(out, **arr)
    print('wrote', out, os.path.getsize(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden'))
    ap.add_argument('cases', nargs='*')
    a = ap.parse_args()
    torch.set_num_threads(8)
    ref_vae, _ = G.import_reference(a.ref)
    if not a.cases or 'B32_C3' in a.cases:
        run_case(ref_vae, 'ref_bench_B32_C3', 32, 3, a.out)
    if not a.cases or 'B64_C8' in a.cases:
        run_case(ref_vae, 'ref_bench_B64_C8', 64, 8, a.out)


if __name__ == '__main__':
    main()
