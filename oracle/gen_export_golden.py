#!/usr/bin/env python3
"""Golden vectors for the post-training exports -- runs the REFERENCE itself (build container only).

TEST INFRASTRUCTURE.  Drives dannyfa/VAE-GAM's own `VAE.plot_GPs` (vae_reg_GP.py:622-689), `VAE.reconstruct` (:585-620) and
`build_model_recons.mk_single_volumes / mk_avg_maps` (build_model_recons.py:15-102) from /root/reference on the CPU, through the
harness stubs of oracle/gen_golden.py plus two that do not touch the arithmetic:
  * nibabel is absent from the image: a capture module stands in whose `save` keeps the array that would have been written
    (as <path>.npy) and whose `load` hands it back, so `mk_avg_maps` re-reads exactly what `reconstruct` produced;
  * `np.float` (build_model_recons.py:74,85) no longer exists in numpy 2: aliased to `float`, its old meaning.
The model is the reference at its seeded initial state (torch.manual_seed(1), the CLI default) -- tests/test_host_logic.py already
pins that this package draws the same initial parameters -- and the noise every forward draws is recorded, so the HIP path can replay it.

Writes tests/golden/export_C8.npz:
  covariates / subjid / vol_num (the data set), noise tape per batch, per-volume statistics of every `recon_<map>` array, statistics +
  a strided sub-sample of every subject-average and grand-average map, and the rows of every `<epoch>_GP_<name>_full.csv`.

Usage:  python oracle/gen_export_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import glob
import os
import sys
import tempfile
import types

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

IMG = (41, 49, 35)
T, BATCH, C, SEED = 12, 4, 8, 31


def capture_nibabel():
    nb = types.ModuleType('nibabel')

    class Nifti1Image:
        def __init__(self, dataobj, affine, header=None):
            self.dataobj, self.affine, self.header = dataobj, affine, header

    def save(img, path):
        np.save(path + '.npy', np.asarray(img.dataobj))

    def load(path):
        if os.path.exists(path + '.npy'):
            return Nifti1Image(np.load(path + '.npy'), np.eye(4), None)
        return Nifti1Image(np.zeros(IMG, np.float32), np.eye(4), None)     # a subject's geometry file: only affine / header are used

    nb.Nifti1Image, nb.save, nb.load = Nifti1Image, save, load
    return nb


def dataset(seed):
    """12 volumes of 2 subjects (7 + 5), wide-range continuous covariates (gen_golden.make_case_inputs' recipe)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cont = rng.normal(size=(T, 6))
    cont[0] = 6.0 + 0.1 * rng.normal(size=6)
    cont[1] = -4.0 + 0.1 * rng.normal(size=6)
    task = (np.arange(T) // 3 % 2 == 0).astype(np.float64)
    subj = (np.arange(T) >= 7).astype(np.int64)
    sex = subj.astype(np.float64)
    vol = np.where(subj == 0, np.arange(T), np.arange(T) - 7)
    df = pd.DataFrame({'subjid': ['s%d' % s for s in subj], 'volume #': vol, 'nii_path': ['ref_s%d.nii' % s for s in subj],
                       'task': task, 'x': cont[:, 0], 'y': cont[:, 1], 'z': cont[:, 2], 'rot_x': cont[:, 3], 'rot_y': cont[:, 4],
                       'rot_z': cont[:, 5], 'sex': sex})
    cov = np.stack([task, *cont.T, sex], 1).astype(np.float32)
    x = np.clip(0.5 + 0.25 * rng.normal(size=(T,) + IMG), 0, 1).astype(np.float32)
    V = int(np.prod(IMG))
    glm = rng.uniform(size=(V, 8)); glm = glm / glm.max(0, keepdims=True)
    return df, cov, x, subj, vol, pd.DataFrame(glm, columns=G.REF_GLM_COLS)


def stats(a, vox):
    a = np.asarray(a, np.float64).ravel()
    return np.concatenate([[a.sum(), (a * a).sum()], a[vox]])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(HERE), 'tests', 'golden'))
    a = ap.parse_args()
    torch.set_num_threads(8)
    import matplotlib
    matplotlib.use('Agg')
    ref_vae, _ = G.import_reference(a.ref)
    nb = capture_nibabel()
    sys.modules['nibabel'] = nb
    ref_vae.nib = nb
    if not hasattr(np, 'float'):
        np.float = float
    import build_model_recons as ref_rec                                  # the reference's module (sys.path set by import_reference)
    ref_rec.nib = nb

    df, cov, x, subj, vol, glm_df = dataset(SEED)
    tmp = tempfile.mkdtemp(prefix='vg_export_')
    train_csv, glm_csv = os.path.join(tmp, 'train.csv'), os.path.join(tmp, 'glm.csv')
    df.to_csv(train_csv); glm_df.to_csv(glm_csv)
    torch.manual_seed(1)
    model = ref_vae.VAE(num_covariates=C, glm_maps=glm_csv, save_dir=tmp, csv_files=[train_csv, train_csv])
    model.epoch = 7
    model.train()                                                          # BatchNorm uses batch statistics in every mode (track_running_stats=False)

    # ---- plot_GPs
    model.plot_GPs(csv_file=train_csv, save_dir=tmp)
    arrays = {'covariates': cov, 'subjid': subj, 'vol_num': vol, 'seed': np.array(SEED), 'epoch': np.array(7), 'batch': np.array(BATCH)}
    names = []
    for f in sorted(glob.glob(os.path.join(tmp, '007_GP_plots', '*_full.csv'))):
        n = os.path.basename(f)[len('007_GP_'):-len('_full.csv')]
        d = pd.read_csv(f)
        names.append(n)
        arrays['gpcsv.%s.index' % n] = d.iloc[:, 0].to_numpy(np.int64)
        arrays['gpcsv.%s.xq' % n] = d['xq'].to_numpy(np.float64)
        arrays['gpcsv.%s.mean' % n] = d['mean'].to_numpy(np.float64)
        arrays['gpcsv.%s.vars' % n] = d['vars'].to_numpy(np.float64)
    arrays['gpcsv.names'] = np.array(names)
    print('plot_GPs:', names)

    # ---- reconstruct / mk_single_volumes / mk_avg_maps with the noise recorded
    tape = G.NoiseTape(SEED + 1000)
    G.patch_noise(tape)
    loader = []
    for s in range(0, T, BATCH):
        sl = slice(s, s + BATCH)
        loader.append({'volume': torch.from_numpy(x[sl]), 'covariates': torch.from_numpy(cov[sl]),
                       'subjid': torch.from_numpy(subj[sl]), 'vol_num': torch.from_numpy(vol[sl])})
    ref_rec.mk_single_volumes(loader, model, train_csv, tmp)
    ref_rec.mk_avg_maps(train_csv, model, tmp, mk_motion_maps=True)
    nb_ = len(loader)
    assert len(tape.draws) == nb_ * (2 + C)
    per = 2 + C
    arrays['eps_w'] = np.concatenate([tape.draws[b * per].numpy() for b in range(nb_)])                 # (T, 1)
    arrays['eps_d'] = np.concatenate([tape.draws[b * per + 1].numpy() for b in range(nb_)])             # (T, 32)
    arrays['eps_beta'] = np.concatenate([torch.stack(tape.draws[b * per + 2:(b + 1) * per]).numpy() for b in range(nb_)], 1)   # (C, T)

    rng = np.random.Generator(np.random.PCG64(7))
    vox = np.sort(rng.choice(int(np.prod(IMG)), 64, replace=False))
    arrays['vox'] = vox
    root = os.path.join(tmp, 'reconstructions', '007_model_recons')
    keys = set()
    for t in range(T):
        d = os.path.join(root, 's%d' % subj[t], 'vol_%d' % vol[t])
        for f in sorted(glob.glob(os.path.join(d, 'recon_*.nii.npy'))):
            k = os.path.basename(f)[len('recon_'):-len('.nii.npy')]
            keys.add(k)
            arr = np.load(f)
            assert arr.shape == IMG
            arrays['vol.%d.%s' % (t, k)] = stats(arr, vox)
    arrays['map_keys'] = np.array(sorted(keys))
    avg_root = os.path.join(tmp, 'reconstructions', '007_avg_model_recons')
    avg_keys = set()
    for f in sorted(glob.glob(os.path.join(avg_root, '**', '*_avg.nii.npy'), recursive=True)):
        rel = os.path.relpath(f, avg_root)[:-len('.nii.npy')]
        arr = np.load(f)
        arrays['avg.%s.stats' % rel] = stats(arr, vox)
        arrays['avg.%s.sub' % rel] = arr[::4, ::4, ::4].astype(np.float64)
        avg_keys.add(rel)
    arrays['avg_keys'] = np.array(sorted(avg_keys))

    # ---- the yardstick: the same maps in FLOAT64 (oracle restatement on the reference model's own parameters, same noise).  A
    # covariate map is gain x decoder output and the reference forms the gain in fp32 through a near-singular B x B Cholesky (SURVEY
    # H2), so how far ITS fp32 maps sit from float64 is the scale on which a second implementation can be asked to agree with them:
    # the test allows 3x this deviation (per statistic, per file) on top of the fp32 floor, as every other conditioning-limited check does.
    import vaegam_oracle as O
    cfg = O.OracleConfig(num_covariates=C, neural_covariates=True)
    p = {'epsilon': model.epsilon.detach().clone()}
    for gname, d_ in model.gp_params.items():
        for k_, v_ in d_.items():
            p['gp.%s.%s' % (gname, k_)] = v_.detach().clone()
    for lname, layer in model._get_layers().items():
        p[lname + '.weight'], p[lname + '.bias'] = layer.weight.detach().clone(), layer.bias.detach().clone()
    glm_full = torch.from_numpy(np.concatenate([np.arange(int(np.prod(IMG)))[:, None].astype(np.float64), glm_df.to_numpy()], 1))
    okeys = ['base'] + [c.name for c in cfg.schema] + ['full_rec']
    fkeys = ['base', 'task', 'x_mot', 'y_mot', 'z_mot', 'pitch_mot', 'roll_mot', 'yaw_mot', 'sex', 'full_rec']   # the reference's file names, :327-342,605
    maps64 = {k: np.zeros((T,) + (int(np.prod(IMG)),)) for k in fkeys}
    for b in range(nb_):
        sl = slice(b * BATCH, (b + 1) * BATCH)
        noise = {'eps_w': tape.draws[b * per], 'eps_d': tape.draws[b * per + 1], 'eps_beta': torch.stack(tape.draws[b * per + 2:(b + 1) * per])}
        p64, x64, c64, n64 = O.to_float64(p, torch.from_numpy(x[sl]), torch.from_numpy(cov[sl]), noise)
        with torch.no_grad():
            out64 = O.forward(p64, cfg, x64, c64, glm_full, n64)
        for ok, fk in zip(okeys, fkeys):
            maps64[fk][sl] = out64['maps'][ok].numpy()

    def dev(ref_stats, m64, sub_ref=None):
        s64 = stats(m64, vox)
        d_ = [abs(ref_stats[0] - s64[0]), abs(ref_stats[1] - s64[1]), np.abs(ref_stats[2:] - s64[2:]).max()]
        if sub_ref is not None:
            d_.append(np.abs(sub_ref - m64.reshape(IMG)[::4, ::4, ::4]).max())
        return np.array(d_)
    for t in range(T):
        for k in sorted(keys):
            arrays['dev.vol.%d.%s' % (t, k)] = dev(arrays['vol.%d.%s' % (t, k)], maps64[k][t])
    for rel in sorted(avg_keys):
        k = os.path.basename(rel)[:-len('_avg')]
        subj_avgs = [maps64[k][subj == s_].mean(0) for s_ in (0, 1)]
        m64 = subj_avgs[int(rel[1])] if rel.startswith('s') and '/' in rel else (subj_avgs[0] + subj_avgs[1]) / 2       # build_model_recons.py:84-91
        arrays['dev.avg.%s' % rel] = dev(arrays['avg.%s.stats' % rel], m64, arrays['avg.%s.sub' % rel])
    worst = max((float(arrays[k_][2]), k_) for k_ in arrays if k_.startswith('dev.vol.'))
    print('largest per-voxel distance of a reference map to float64: %.3g (%s)' % worst)
    print('maps per volume:', sorted(keys)); print('average maps:', sorted(avg_keys))
    np.savez_compressed(os.path.join(a.out, 'export_C8.npz'), **arrays)
    print('wrote', os.path.join(a.out, 'export_C8.npz'), os.path.getsize(os.path.join(a.out, 'export_C8.npz')), 'bytes')


if __name__ == '__main__':
    main()
