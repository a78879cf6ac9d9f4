"""Seeded synthetic checker-control data set (restates the geometry and time course of
the reference's add_control_signal.py:105-130,146 and utils.py:93-123,170-178 -- those scripts need
real fMRI and an MNIST download, neither of which exists offline).

Everything is generated from numpy PCG64 seeds: volumes in [0,1] (already divided by the
reference's 3284.5), a hand-authored 13x13 'Large3' glyph x 10 slices at [15:25, 34:47, 9:22]
switched by the control block design, z-scored wide-range motion covariates, max-scaled GLM maps.
"""
import numpy as np

from . import utils

# hand-authored 13x13 binary '3' (the reference thresholds a resized MNIST '3'; MNIST is not fetchable here)
_GLYPH = np.array([[0, 0, 0, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0],
                   [0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0],
                   [0, 0, 1, 0, 0, 0, 0, 0, 1, 1, 1, 0, 0],
                   [0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0],
                   [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 0, 0],
                   [0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 0, 0, 0],
                   [0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 0, 0, 0],
                   [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 0, 0],
                   [0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0],
                   [0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0],
                   [0, 0, 1, 0, 0, 0, 0, 0, 1, 1, 1, 0, 0],
                   [0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0],
                   [0, 0, 0, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0]], dtype=np.float64)


def large3_signal(img_shape=(41, 49, 35), intensity=1000.0):
    """Control-signal volume: glyph rotated -90 deg, broadcast to 10 slices, at [15:25, 34:47, 9:22]
    (x2 per axis at 82x98x70), amplitude intensity/3284.5."""
    scale = img_shape[0] // 41
    glyph = np.rot90(_GLYPH, k=-1)                       # ndimage.rotate(sig, -90) on a square binary image
    sig = np.broadcast_to(glyph, (10, 13, 13)).copy()
    if scale > 1:
        sig = sig.repeat(scale, 0).repeat(scale, 1).repeat(scale, 2)
    out = np.zeros(img_shape)
    out[15 * scale:25 * scale, 34 * scale:47 * scale, 9 * scale:22 * scale] += sig
    return out * (intensity / 3284.5)


def make_dataset(num_subjects=2, vols_per_subject=98, num_covariates=8, img_shape=(41, 49, 35), seed=0, dtype=np.float32):
    """Returns dict(volumes (N,X,Y,Z), covariates (N,C), subjid (N,), vol_num (N,), xu_ranges, glm (V,C+1), names).
    Covariate columns follow the reference layout [task, x, y, z, rot_x, rot_y, rot_z, sex] (C=8), its
    first C columns for C<8, and task + (C-8) extra continuous + 6 motion + sex for C>8."""
    rng = np.random.Generator(np.random.PCG64(seed))
    S, T, C = num_subjects, vols_per_subject, num_covariates
    N = S * T
    X, Y, Z = img_shape
    gx, gy, gz = np.meshgrid(np.linspace(-1, 1, X), np.linspace(-1, 1, Y), np.linspace(-1, 1, Z), indexing='ij')
    brain = ((gx / 0.85) ** 2 + (gy / 0.9) ** 2 + (gz / 0.8) ** 2) <= 1.0
    task = utils.control_stimulus_to_neural(np.arange(1, T + 1) * 1.4).astype(np.float64)      # add_control_signal.py:120-123
    sig = large3_signal(img_shape)
    vols = np.empty((N, X, Y, Z), dtype=dtype)
    for s in range(S):
        smooth = 0.5 + 0.3 * np.sin(2.0 * gx + rng.uniform(0, 3)) * np.cos(1.5 * gy + rng.uniform(0, 3)) * np.cos(gz + rng.uniform(0, 3))
        base = brain * smooth
        for t in range(T):
            v = base + sig * task[t] + 0.01 * rng.standard_normal((X, Y, Z))
            vols[s * T + t] = np.clip(v, 0.0, 1.0)
    ncont = 6 if C <= 8 else C - 2
    cont = utils.zscore_columns(rng.standard_normal((N, ncont)))
    cont[0] = 6.0 + 0.1 * rng.standard_normal(ncont)            # two outliers per column: every covariate spans ~[-4, 6],
    cont[1] = -4.0 + 0.1 * rng.standard_normal(ncont)           # which keeps the inducing-point kernel Ku well conditioned
    sex = np.repeat((np.arange(S) % 2).astype(np.float64), T)
    full = np.concatenate([np.tile(task, S)[:, None], cont, sex[:, None]], 1)
    covariates = full[:, :C] if C <= 8 else full
    motion = cont[:, -6:] if C > 8 else cont
    xu_all = [[float(cont[:, j].min()) - 1e-3, float(cont[:, j].max()) + 1e-3] for j in range(ncont)]
    glm = rng.uniform(size=(max(C, 8), X * Y * Z))
    glm = utils.scale_beta_maps(glm).T                          # (V, C)
    glm = np.concatenate([np.arange(X * Y * Z, dtype=np.float64)[:, None], glm], 1)
    return dict(volumes=vols, covariates=covariates.astype(dtype), subjid=np.repeat(np.arange(S), T).astype(np.int64),
                vol_num=np.tile(np.arange(T), S), xu_ranges=xu_all, glm=glm, motion=motion, task=task)


def write_csvs(ds, out_dir, prefix='synth'):
    """Write the data set in the reference's on-disk layout: one .npy 4-D file per subject, a
    per-volume CSV (pre_proc_vaefmri.py:126-127 columns) and the GLM-map CSV (70315 x C + index)."""
    import os
    import pandas as pd
    os.makedirs(out_dir, exist_ok=True)
    S = int(ds['subjid'].max()) + 1
    T = len(ds['subjid']) // S
    rows = []
    for s in range(S):
        path = os.path.join(out_dir, '%s_subj%02d.npy' % (prefix, s))
        np.save(path, np.moveaxis(ds['volumes'][s * T:(s + 1) * T], 0, -1) * 3284.5)
        for t in range(T):
            c = ds['covariates'][s * T + t]
            c8 = np.zeros(8); c8[:min(8, len(c))] = c[:8]
            rows.append(['subj%02d' % s, t, path] + c8.tolist())
    cols = ['subjid', 'volume #', 'nii_path', 'task', 'x', 'y', 'z', 'rot_x', 'rot_y', 'rot_z', 'sex']
    df = pd.DataFrame(rows, columns=cols)
    csv = os.path.join(out_dir, prefix + '.csv')
    df.to_csv(csv)
    glm_csv = os.path.join(out_dir, prefix + '_glm.csv')
    pd.DataFrame(ds['glm'][:, 1:9], columns=cols[3:]).to_csv(glm_csv)
    return csv, glm_csv
