#!/usr/bin/env python3
"""Golden vectors for the real-data INPUT path -- runs the REFERENCE'S OWN SCRIPTS (build container only).

TEST INFRASTRUCTURE.  `get_beta_map_regularizer.py` (the OLS maps the model's GLM regulariser reads, :73-107) and
`pre_proc_vaefmri.py` (the per-volume CSV `FMRIDataset` reads, :66-133) are stand-alone argparse scripts over fmriprep / FSL
directory trees.  This generator builds a small synthetic tree in their layout from a seeded recipe (`recipe()` below), runs the two
scripts from /root/reference with runpy (never copied, never edited; nibabel -- absent from the image -- stood in for by a module whose
`load` hands back the array stored beside the path, tensorboard by gen_golden's no-op stub) and stores what they wrote:
  scld_GLM_beta_maps.csv            -> glm.columns, glm.index, glm.values (V x 8)
  preproc_dset_zscored_<date>_..csv -> csv.columns, csv.index, csv.subjid, csv.volume, csv.nii_path, csv.values (N x 8: task, 6 motion, sex)
tests/test_host_logic.py rebuilds the inputs from the recipe and holds vae_gam_amd.utils.{read_design_mat, glm_beta_maps,
preproc_table} to these outputs.   Usage:  python oracle/gen_preproc_golden.py [--ref /root/reference]
"""
import argparse
import glob
import os
import runpy
import sys
import tempfile
import types

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

DIMS = (6, 5, 4, 24)           # x, y, z, time
SUBJECTS = ['sub-A00010001', 'sub-A00010002', 'sub-A00010003']
SEED = 41


def recipe(seed=SEED):
    """Per subject: filtered 4-D data, the FSL design matrix (task + 2 nuisance + 6 motion columns), fmriprep motion regressors, sex;
    plus the group-level sex cope map."""
    rng = np.random.Generator(np.random.PCG64(seed))
    X, Y, Z, T = DIMS
    out = {}
    for k, s in enumerate(SUBJECTS):
        task = ((np.arange(T) // 4) % 2 == 0).astype(np.float64)
        design = np.concatenate([task[:, None] - 0.5, rng.normal(size=(T, 2)), 0.3 * rng.normal(size=(T, 6))], 1)
        data = 100.0 + 10.0 * rng.normal(size=(X, Y, Z, T)) + 5.0 * task
        motion = rng.normal(size=(T, 6)) * np.array([0.3, 0.2, 0.4, 0.01, 0.02, 0.015]) + 0.1 * k
        out[s] = dict(design=np.round(design, 6), data=data, motion=motion, sex=k % 2)
    out['sex_map'] = rng.normal(size=(X, Y, Z))
    return out


def write_design_mat(path, m):
    """FSL's design.mat text layout: 5 header lines, then tab-separated rows with a trailing tab (utils.read_design_mat skips 5 lines)."""
    with open(path, 'w') as f:
        f.write('/NumWaves\t%d\n/NumPoints\t%d\n/PPheights\t%s\n\n/Matrix\n' % (m.shape[1], m.shape[0], '\t'.join('%e' % v for v in np.ptp(m, axis=0))))
        for row in m:
            f.write('\t'.join('%e' % v for v in row) + '\t\n')


def capture_nibabel():
    nb = types.ModuleType('nibabel')

    class Img:
        def __init__(self, arr):
            self.dataobj = arr

    nb.load = lambda path: Img(np.load(str(path) + '.npy'))
    return nb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden'))
    a = ap.parse_args()
    G.install_stubs()
    sys.modules['nibabel'] = capture_nibabel()
    sys.path.insert(0, a.ref)
    rc = recipe()
    tmp = tempfile.mkdtemp(prefix='vg_preproc_')
    root = os.path.join(tmp, 'data'); os.makedirs(root)
    sex_rows = []
    for s in SUBJECTS:
        feat = os.path.join(root, s, 'func', 'run1_corrected.feat'); os.makedirs(feat)
        open(os.path.join(feat, 'filtered_func_data.nii.gz'), 'w').close()
        np.save(os.path.join(feat, 'filtered_func_data.nii.gz.npy'), rc[s]['data'])
        write_design_mat(os.path.join(feat, 'design.mat'), rc[s]['design'])
        nii = os.path.join(root, s, 'func', s + '_preproc_bold_brainmasked_resampled.nii.gz')
        open(nii, 'w').close(); np.save(nii + '.npy', rc[s]['data'])
        tsv = os.path.join(root, s, 'func', s + '_task-CHECKERBOARD_acq-1400_desc-confounds_regressors_v1.tsv')
        pd.DataFrame(np.concatenate([rc[s]['motion'], np.zeros((DIMS[3], 1))], 1),
                     columns=['trans_x', 'trans_y', 'trans_z', 'rot_x', 'rot_y', 'rot_z', 'csf']).to_csv(tsv, sep='\t', index=False)
        sex_rows.append((s, rc[s]['sex']))
    sex_map = os.path.join(tmp, 'sex_cope.nii.gz'); open(sex_map, 'w').close(); np.save(sex_map + '.npy', rc['sex_map'])
    sex_csv = os.path.join(tmp, 'sex.csv')
    pd.DataFrame(sex_rows, columns=['subjID', 'gender ']).to_csv(sex_csv, index=False)       # the reference reads the column 'gender ' (sic, :101)

    out_glm = os.path.join(tmp, 'glm'); out_csv = os.path.join(tmp, 'csv')
    argv0 = list(sys.argv)
    try:
        sys.argv = ['get_beta_map_regularizer.py', '--root_dir', root, '--output_dir', out_glm, '--data_dims'] + [str(v) for v in DIMS] + ['--sex_covars_map', sex_map]
        runpy.run_path(os.path.join(a.ref, 'get_beta_map_regularizer.py'), run_name='__main__')
        sys.argv = ['pre_proc_vaefmri.py', '--data_dir', root, '--save_dir', out_csv, '--control', 'True', '--control_int', '1000', '--set_tag', 'TRAIN',
                    '--sex_info', sex_csv]
        runpy.run_path(os.path.join(a.ref, 'pre_proc_vaefmri.py'), run_name='__main__')
    finally:
        sys.argv = argv0
    glm = pd.read_csv(os.path.join(out_glm, 'scld_GLM_beta_maps.csv'))
    csvs = glob.glob(os.path.join(out_csv, 'preproc_dset_zscored_*_TRAIN_large3_1000_control_simple_ts.csv'))
    assert len(csvs) == 1, csvs
    df = pd.read_csv(csvs[0])
    arr = {'seed': np.array(SEED), 'dims': np.array(DIMS), 'subjects': np.array(SUBJECTS),
           'glm.columns': np.array(list(glm.columns)), 'glm.index': glm.iloc[:, 0].to_numpy(np.int64), 'glm.values': glm.iloc[:, 1:].to_numpy(np.float64),
           'csv.columns': np.array(list(df.columns)), 'csv.index': df.iloc[:, 0].to_numpy(np.int64), 'csv.subjid': df['subjid'].to_numpy(str),
           'csv.volume': df['volume #'].to_numpy(np.int64), 'csv.nii_name': np.array([os.path.basename(p) for p in df['nii_path']]),
           'csv.values': df[['task', 'x', 'y', 'z', 'rot_x', 'rot_y', 'rot_z', 'sex']].to_numpy(np.float64),
           'csv.name_suffix': np.array(os.path.basename(csvs[0]).split('_', 6)[-1])}
    out = os.path.join(a.out, 'preproc_ref.npz')
    np.savez_compressed(out, **arr)
    print('subject order the scripts used:', list(dict.fromkeys(df['subjid'])))
    print('wrote', out, os.path.getsize(out), 'bytes;  glm', arr['glm.values'].shape, 'csv', arr['csv.values'].shape)


if __name__ == '__main__':
    main()
