"""Host-side helpers of the hot path (drop-in for the reference's utils.py:22-73) plus the
pieces of its preprocessing the synthetic generator restates (utils.py:93-123, 170-178)."""
import argparse

import numpy as np


def hrf(times):
    """Double-gamma haemodynamic response, peak-normalised to 0.6 (utils.py:22-36)."""
    from scipy.stats import gamma
    times = np.asarray(times, dtype=np.float64)
    values = gamma.pdf(times, 6) - 0.35 * gamma.pdf(times, 12)
    return values / np.max(values) * 0.6


def get_xu_ranges(csv_files, eps=1e-3):
    """[min-eps, max+eps] of each motion regressor over train and test CSVs (utils.py:39-56)."""
    import pandas as pd
    frames = [pd.read_csv(f) for f in csv_files[:2]]
    out = []
    for reg in ['x', 'y', 'z', 'rot_x', 'rot_y', 'rot_z']:
        lo = min(float(df[reg].min()) for df in frames)
        hi = max(float(df[reg].max()) for df in frames)
        out.append([lo - eps, hi + eps])
    return out


def str2bool(v):
    """argparse bool flag parser (utils.py:59-73; the reference forgets to import argparse)."""
    if isinstance(v, bool):
        return v
    s = str(v).lower()
    if s in ('yes', 'true', 't', 'y', '1'):
        return True
    if s in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


def control_stimulus_to_neural(vol_times, block=20.0):
    """Block design of the control experiments: ON during even 20-s blocks, starting ON (utils.py:93-111)."""
    t = np.asarray(vol_times) // block
    return (t.astype(np.int64) % 2 == 0).astype(np.int64)


def zscore_columns(a):
    """Column-wise z-score with population std (utils.py:113-123)."""
    a = np.asarray(a, dtype=np.float64)
    return (a - a.mean(0)) / a.std(0)


def scale_beta_maps(beta_maps):
    """Divide each map (row) by its maximum (utils.py:170-178)."""
    beta_maps = np.array(beta_maps, dtype=np.float64, copy=True)
    return beta_maps / beta_maps.max(1, keepdims=True)


def glm_beta_maps(design, data, sex_map=None):
    """Least-squares GLM maps used as the model's regulariser targets (get_beta_map_regularizer.py:94-107):
    design (T, R) stacked design matrices (task + 6 motion columns), data (V, T) filtered voxel time courses
    -> (R [+1], V) maps beta = (G^T G)^-1 G^T Y, the optional per-voxel `sex_map` appended as a last row, each row
    divided by its maximum (scale_beta_maps).  The FSL .feat directory walk around it is preprocessing and not part
    of this package; the `glm_maps` CSV the model reads is `pd.DataFrame(maps.T, columns=[...]).to_csv(...)`."""
    G = np.asarray(design, dtype=np.float64)
    Y = np.asarray(data, dtype=np.float64)
    if Y.shape[1] != G.shape[0]:
        raise ValueError('data has %d time points, design matrix %d rows' % (Y.shape[1], G.shape[0]))
    pinv = np.linalg.inv(G.T @ G) @ G.T                       # (R, T), as the reference forms it (:95-96)
    beta = pinv @ Y.T                                         # (R, V)
    if sex_map is not None:
        beta = np.concatenate([beta, np.asarray(sex_map, dtype=np.float64).reshape(1, -1)], axis=0)
    return scale_beta_maps(beta)


def read_design_mat(mat_file_path):
    """FSL feat `design.mat` -> (time points, regressors) array (utils.py:153-168): five header lines, then tab-separated rows."""
    import re
    rows = []
    with open(mat_file_path) as f:
        for line in f.readlines()[5:]:
            cells = [c for c in re.split(r'\t+', line.rstrip()) if c != '']
            if cells:
                rows.append([float(c) for c in cells])
    return np.array(rows)


def glm_design_columns(design_mat):
    """The columns of one subject's FSL design matrix the GLM regulariser uses (get_beta_map_regularizer.py:86-90): the task regressor
    (first column) and the six motion parameters (last six)."""
    m = np.asarray(design_mat, dtype=np.float64)
    return np.concatenate([m[:, :1], m[:, -6:]], axis=1)


PREPROC_COLUMNS = ["subjid", "volume #", "nii_path", "task", "x", "y", "z", "rot_x", "rot_y", "rot_z", "sex"]


def stimulus_to_neural(vol_times, block=20.0):
    """Block design of the checkerboard task: OFF first, ON during odd 20-s blocks (utils.py:75-91)."""
    t = (np.asarray(vol_times) // block).astype(np.int64)
    return (t % 2 != 0).astype(np.int64)


def preproc_table(subjects, control=False, tr=1.4):
    """The per-volume table `FMRIDataset` reads, as pre_proc_vaefmri.py:97-129 builds it: one row per (subject, volume) with the columns
    PREPROC_COLUMNS -- task = the block time course at (volume+1)*TR (`control`: the control experiments' ON-first design), the six
    fmriprep motion regressors, the subject's sex -- and the motion columns z-scored over ALL rows with the population standard
    deviation (utils.py:113-123).  `subjects`: iterable of dicts {subjid, nii_path, motion (T,6) [trans_x, trans_y, trans_z, rot_x,
    rot_y, rot_z], sex}.  Returns a pandas DataFrame; `.to_csv(path)` gives the reference's file (index column first)."""
    import pandas as pd
    rows = []
    for s in subjects:
        mot = np.asarray(s['motion'], dtype=np.float64)
        T = mot.shape[0]
        times = np.arange(1, T + 1) * tr
        neural = control_stimulus_to_neural(times) if control else stimulus_to_neural(times)
        for v in range(T):
            rows.append((s['subjid'], v, s['nii_path'], neural[v]) + tuple(mot[v]) + (s['sex'],))
    df = pd.DataFrame(rows, columns=PREPROC_COLUMNS)
    cols = ['x', 'y', 'z', 'rot_x', 'rot_y', 'rot_z']
    df[cols] = zscore_columns(df[cols].to_numpy())
    return df


def log_map(writer, img_shape, maps, slice_idx, tag, batch_size, log_type):
    """Axial slice of every batch element to the writer (subset of utils.py:373-389; opt-in)."""
    add = getattr(writer, 'add_images', None)
    if add is None:
        return
    try:
        import torch
        m = maps.detach().reshape(batch_size, *img_shape)[:, :, :, slice_idx] if isinstance(maps, torch.Tensor) \
            else np.asarray(maps).reshape(batch_size, *img_shape)[:, :, :, slice_idx]
        add('%s_%s_slice_%d' % (tag, log_type, slice_idx), m[:, None], dataformats='NCHW')
    except Exception:
        pass
