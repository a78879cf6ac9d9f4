"""Single-volume, subject-average and grand-average reconstruction maps (reference: build_model_recons.py:15-116).

`mk_single_volumes` keeps the reference's interface and directory layout (one `recon_<map>.nii` per volume).
`mk_avg_maps` produces the same `<map>_avg.nii` files, but from sums that `VAE.reconstruct` accumulated ON THE
DEVICE while it made the single volumes (`model.recon_sums`), instead of re-reading every per-volume file the
way the reference does; called without a preceding `mk_single_volumes(..., write_volumes=...)` pass it runs the
loader itself with file writing off.
"""
import os

import numpy as np
import pandas as pd

from . import nifti

MAPS_ALL = ['base', 'task', 'full_rec', 'x_mot', 'y_mot', 'z_mot', 'pitch_mot', 'roll_mot', 'yaw_mot', 'sex']   # :66-67


def _subjects(csv_file):
    dset = pd.read_csv(csv_file)
    return dset.iloc[:, 1].unique().tolist(), dset.iloc[:, 3].unique().tolist()          # subjid, nii_path (positional, as DataClass_GP)


def mk_single_volumes(loader, model, csv_file, save_dir, write_volumes=True, noise=None):
    subjs, ref_niis = _subjects(csv_file)
    ckpt_num = str(model.epoch).zfill(3)
    subj_dirs = []
    for s in subjs:
        d = os.path.join(save_dir, 'reconstructions', '{}_model_recons'.format(ckpt_num), str(s))
        os.makedirs(d, exist_ok=True)
        subj_dirs.append(d)
    model.reconstruct(loader, ref_niis, subj_dirs, write_volumes=write_volumes, noise=noise)


def mk_avg_maps(csv_file, model, save_dir, mk_motion_maps=False, loader=None):
    subjs, ref_niis = _subjects(csv_file)
    if getattr(model, 'recon_sums', None) is None:
        if loader is None:
            raise ValueError('mk_avg_maps: run mk_single_volumes first or pass the loader')
        mk_single_volumes(loader, model, csv_file, save_dir, write_volumes=False)
    ckpt_num = str(model.epoch).zfill(3)
    avg_dir = os.path.join(save_dir, 'reconstructions', '{}_avg_model_recons'.format(ckpt_num))
    os.makedirs(avg_dir, exist_ok=True)
    sums, counts = model.recon_sums                          # {map: (S, V) device tensor}, (S,) device tensor
    keys = list(sums.keys())
    maps = [m for m in MAPS_ALL if m in keys] + [k for k in keys if k not in MAPS_ALL]
    if not mk_motion_maps:
        maps = [m for m in maps if m in ('base', 'task', 'full_rec', 'sex')]              # :68-70
    cnt = counts.clamp_min(1).unsqueeze(1)
    out = {}
    shape = tuple(model.img_shape)
    for m in maps:
        subj_avg = (sums[m] / cnt).cpu().numpy().astype(np.float64)                      # (S, V)
        for i, s in enumerate(subjs):
            d = os.path.join(avg_dir, str(s)); os.makedirs(d, exist_ok=True)
            ref = ref_niis[i] if (i < len(ref_niis) and str(ref_niis[i]).endswith(('.nii', '.nii.gz')) and os.path.exists(str(ref_niis[i]))) else None
            nifti.write_nifti1(os.path.join(d, '{}_avg.nii'.format(m)), subj_avg[i].reshape(shape), ref)
        grand = subj_avg[:len(subjs)].mean(0)                                            # mean of subject means, :96
        ref0 = ref_niis[0] if (ref_niis and str(ref_niis[0]).endswith(('.nii', '.nii.gz')) and os.path.exists(str(ref_niis[0]))) else None
        nifti.write_nifti1(os.path.join(avg_dir, '{}_avg.nii'.format(m)), grand.reshape(shape), ref0)
        out[m] = grand.reshape(shape)
    return out
