"""Post-training exports against vectors produced by the REFERENCE's own plot_GPs, reconstruct,
mk_single_volumes and mk_avg_maps (oracle/gen_export_golden.py ran them in the build container; tests/golden/export_C8.npz).
The data set is a seeded recipe shared with the generator; the noise every reference forward drew is replayed."""
import os
import sys

import numpy as np
import pandas as pd
import pytest
import torch

import vae_gam_amd  # noqa: F401
from vae_gam_amd import DataClass_GP, build_model_recons as R
from vae_gam_amd.vae_reg_GP import VAE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
GOLD = os.path.join(ROOT, 'tests', 'golden', 'export_C8.npz')


def _setup(tmp_path, device):
    import gen_export_golden as E                                       # the recipe only: nothing of the reference is imported
    g = np.load(GOLD)
    df, cov, x, subj, vol, glm_df = E.dataset(int(g['seed']))
    np.testing.assert_array_equal(cov, g['covariates'])
    train_csv, glm_csv = str(tmp_path / 'train.csv'), str(tmp_path / 'glm.csv')
    df.to_csv(train_csv); glm_df.to_csv(glm_csv)
    torch.manual_seed(1)
    m = VAE(num_covariates=8, glm_maps=glm_csv, save_dir=str(tmp_path), csv_files=[train_csv, train_csv], device_name=device)
    m.epoch = int(g['epoch'])
    return g, m, train_csv, (cov, x, subj, vol)


def _check_gp_csvs(g, tmp_path):
    names = [str(n) for n in g['gpcsv.names']]
    assert sorted(names) == sorted(['x', 'y', 'z', 'xrot', 'yrot', 'zrot'])
    for n in names:
        f = pd.read_csv(str(tmp_path / '007_GP_plots' / ('007_GP_%s_full.csv' % n)))
        assert list(f.columns[1:]) == ['xq', 'mean', 'vars']
        np.testing.assert_array_equal(f.iloc[:, 0].to_numpy(np.int64), g['gpcsv.%s.index' % n])         # pandas' sort order of the rows
        # xq: the reference keeps the CSV's float64, the data path here rounds to fp32 first (DataClass_GP hands fp32 covariates)
        np.testing.assert_allclose(f['xq'].to_numpy(), g['gpcsv.%s.xq' % n], rtol=1e-6, atol=1e-7)
        # the reference evaluates its posterior in mixed fp32/fp64; 1e-4 absolute is SURVEY 8c's tolerance for f_bar / Sigma
        np.testing.assert_allclose(f['mean'].to_numpy(), g['gpcsv.%s.mean' % n], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(f['vars'].to_numpy(), g['gpcsv.%s.vars' % n], rtol=1e-4, atol=1e-4)


def test_plot_GPs_csvs_match_reference_cpu(tmp_path):
    g, m, train_csv, _ = _setup(tmp_path, 'cpu')
    m.plot_GPs(csv_file=train_csv, save_dir=str(tmp_path))
    _check_gp_csvs(g, tmp_path)


@pytest.mark.gpu
def test_exports_match_reference(tmp_path):
    g, m, train_csv, (cov, x, subj, vol) = _setup(tmp_path, 'cuda')
    m.plot_GPs(csv_file=train_csv, save_dir=str(tmp_path))
    _check_gp_csvs(g, tmp_path)

    Bt = int(g['batch']); T = cov.shape[0]
    loader = [{'volume': torch.from_numpy(x[s:s + Bt]), 'covariates': torch.from_numpy(cov[s:s + Bt]),
               'subjid': torch.from_numpy(subj[s:s + Bt]), 'vol_num': torch.from_numpy(vol[s:s + Bt])} for s in range(0, T, Bt)]

    def noise(bi, B):
        sl = slice(bi * Bt, bi * Bt + B)
        return {'eps_w': torch.from_numpy(g['eps_w'][sl]).cuda(), 'eps_d': torch.from_numpy(g['eps_d'][sl]).cuda(),
                'eps_beta': torch.from_numpy(np.ascontiguousarray(g['eps_beta'][:, sl])).cuda()}
    R.mk_single_volumes(loader, m, train_csv, str(tmp_path), noise=noise)
    R.mk_avg_maps(train_csv, m, str(tmp_path), mk_motion_maps=True)

    vox = g['vox']

    def check(path, want, what, dev):
        a = DataClass_GP.read_nifti1(path).astype(np.float64).ravel()
        got = np.concatenate([[a.sum(), (a * a).sum()], a[vox]])
        # Bands = the fp32 floor of SURVEY 8c + 3x the distance of THE REFERENCE'S OWN fp32 file to the same map in float64, measured per
        # file and statistic by the generator (`dev.*` in the fixture: [|sum|, |sum of squares|, max per voxel(, max over the strided
        # sub-sample)]).  A covariate map is gain x decoder output and the reference draws the gain in fp32 through a near-singular B x B
        # Cholesky -- up to 1.1e-4 per voxel from float64 on vol 11 / pitch_mot -- while the gain block here runs in fp64;
        # the same rule as every other conditioning-limited comparison in tests/.  Floors: maps 1e-5 per voxel (+ the fp32 file round
        # trip), the signed sum by the Cauchy-Schwarz bound 1e-5 * sqrt(sum of squares * V), the sum of squares rel 1e-4.
        np.testing.assert_allclose(got[2:], want[2:], rtol=1e-5, atol=2e-5 + 3 * dev[2], err_msg=what)
        np.testing.assert_allclose(got[0], want[0], rtol=1e-5, atol=1e-5 * np.sqrt(want[1] * a.size) + 3 * dev[0], err_msg=what + ' (sum)')
        np.testing.assert_allclose(got[1], want[1], rtol=1e-4, atol=3 * dev[1], err_msg=what + ' (sum of squares)')
    root = tmp_path / 'reconstructions' / '007_model_recons'
    keys = [str(k) for k in g['map_keys']]
    for t in range(T):
        d = root / ('s%d' % subj[t]) / ('vol_%d' % vol[t])
        assert sorted(os.listdir(d)) == sorted('recon_%s.nii' % k for k in keys)
        for k in keys:
            check(str(d / ('recon_%s.nii' % k)), g['vol.%d.%s' % (t, k)], 'vol %d %s' % (t, k), g['dev.vol.%d.%s' % (t, k)])
    avg_root = tmp_path / 'reconstructions' / '007_avg_model_recons'
    avg_keys = [str(k) for k in g['avg_keys']]
    found = sorted(os.path.relpath(os.path.join(dp, f), str(avg_root))[:-4] for dp, _, fs in os.walk(str(avg_root)) for f in fs)
    assert found == sorted(avg_keys)
    for k in avg_keys:
        check(str(avg_root / (k + '.nii')), g['avg.%s.stats' % k], 'avg ' + k, g['dev.avg.%s' % k])
        a = DataClass_GP.read_nifti1(str(avg_root / (k + '.nii')))
        np.testing.assert_allclose(a[::4, ::4, ::4], g['avg.%s.sub' % k], rtol=1e-5, atol=2e-5 + 3 * g['dev.avg.%s' % k][3], err_msg='avg ' + k)
