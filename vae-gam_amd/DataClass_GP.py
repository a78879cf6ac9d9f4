"""Dataset / loaders for the VAE-GAM (drop-in surface of the reference's DataClass_GP.py).

Same CSV layout (positional columns: index, subjid, volume #, nii_path, task, x, y, z, rot_x,
rot_y, rot_z, sex; DataClass_GP.py:31-46), same sample dictionary (DataClass_GP.py:63-71) and
same loader dictionary (DataClass_GP.py:73-89).  Differences that matter on an MI355X node:
each 4-D file is decoded ONCE and kept (the reference re-reads the whole 4-D NIfTI for every
sample, DataClass_GP.py:48), `.npy` volumes are memory-mapped, and `DeviceResidentData` keeps a
whole synthetic or pre-loaded set in HBM so the train loop never touches the host.
"""
import gzip
import os
import struct

import numpy as np
import pandas as pd
import torch
from torch.utils.data import DataLoader, Dataset

GLOBAL_MAX = 3284.5       # DataClass_GP.py:49


def read_nifti1(path):
    """Minimal NIfTI-1 reader (.nii / .nii.gz, single file): returns the data array in file
    order (Fortran layout -> shape dim[1..ndim]) with scl_slope/scl_inter applied.  nibabel is
    not available in the image; the reference goes through nib.load(...).dataobj."""
    opener = gzip.open if path.endswith('.gz') else open
    with opener(path, 'rb') as f:
        raw = f.read()
    if struct.unpack('<i', raw[:4])[0] == 348:
        en = '<'
    elif struct.unpack('>i', raw[:4])[0] == 348:
        en = '>'
    else:
        raise ValueError('%s: not a NIfTI-1 file' % path)
    dim = struct.unpack(en + '8h', raw[40:56])
    datatype, bitpix = struct.unpack(en + '2h', raw[70:74])
    vox_offset = int(struct.unpack(en + 'f', raw[108:112])[0])
    slope, inter = struct.unpack(en + '2f', raw[112:120])
    dtypes = {2: 'u1', 4: 'i2', 8: 'i4', 16: 'f4', 64: 'f8', 256: 'i1', 512: 'u2', 768: 'u4'}
    if datatype not in dtypes:
        raise ValueError('%s: unsupported NIfTI datatype %d' % (path, datatype))
    shape = tuple(int(d) for d in dim[1:1 + dim[0]])
    n = int(np.prod(shape))
    a = np.frombuffer(raw, dtype=np.dtype(en + dtypes[datatype]), count=n, offset=max(vox_offset, 352))
    a = a.reshape(shape, order='F')
    if slope not in (0.0, 1.0) or inter != 0.0:
        if slope != 0.0:
            a = a * slope + inter
    return a


_VOLUME_CACHE = {}


def load_4d(path):
    """4-D array (X,Y,Z,T) of a subject, decoded once per process."""
    if path not in _VOLUME_CACHE:
        if path.endswith('.npy'):
            _VOLUME_CACHE[path] = np.load(path, mmap_mode='r')
        else:
            _VOLUME_CACHE[path] = read_nifti1(path)
    return _VOLUME_CACHE[path]


class FMRIDataset(Dataset):
    """CSV-indexed fMRI volumes (reference DataClass_GP.py:11-60)."""

    def __init__(self, csv_file, transform=None):
        self.df = pd.read_csv(csv_file)
        self.transform = transform
        self._subjects = self.df.subjid.unique().tolist()      # hoisted out of __getitem__ (DataClass_GP.py:31)

    def __len__(self):
        return len(self.df)

    def __getitem__(self, idx):
        row = self.df.iloc[idx]
        subj = row.iloc[1]
        vol_num = row.iloc[2]
        fmri = load_4d(row.iloc[3])
        volume = np.asarray(fmri[:, :, :, int(vol_num)])
        scld_vol = np.true_divide(volume.flatten(), GLOBAL_MAX).reshape(volume.shape)
        sample = {'subj_idx': self._subjects.index(subj), 'subj': subj, 'volume': scld_vol, 'vol_num': vol_num,
                  'task': row.iloc[4], 'trans_x': row.iloc[5], 'trans_y': row.iloc[6], 'trans_z': row.iloc[7],
                  'rot_x': row.iloc[8], 'rot_y': row.iloc[9], 'rot_z': row.iloc[10], 'sex': row.iloc[11]}
        if self.transform:
            sample = self.transform(sample)
        return sample


class ToTensor(object):
    """Sample dict -> tensors the model consumes (DataClass_GP.py:62-71)."""

    def __call__(self, sample):
        covars = np.array([sample['task'], sample['trans_x'], sample['trans_y'], sample['trans_z'], sample['rot_x'],
                           sample['rot_y'], sample['rot_z'], sample['sex']], dtype=np.float64)
        return {'covariates': torch.from_numpy(covars).float(),
                'volume': torch.from_numpy(np.ascontiguousarray(sample['volume'])).float(),
                'subjid': torch.tensor(sample['subj_idx'], dtype=torch.int64),
                'vol_num': torch.tensor(sample['vol_num'], dtype=torch.float64)}


def setup_data_loaders(batch_size=32, shuffle=(True, False, False), train_csv='', test_csv='', prefetch_device=None):
    """{'Shuffled_train', 'UnShuffled_train', 'test'} loaders (DataClass_GP.py:73-89).
    prefetch_device (extension): a CUDA device -> the loaders collate into pinned host memory and are wrapped in DevicePrefetcher."""
    train_dataset = FMRIDataset(csv_file=train_csv, transform=ToTensor())
    test_dataset = FMRIDataset(csv_file=test_csv, transform=ToTensor())
    pin = prefetch_device is not None and torch.device(prefetch_device).type == 'cuda'
    mk = lambda ds, sh: DataLoader(ds, batch_size=batch_size, shuffle=sh, num_workers=0, pin_memory=pin)
    out = {'Shuffled_train': mk(train_dataset, shuffle[0]), 'UnShuffled_train': mk(train_dataset, shuffle[1]),
           'test': mk(test_dataset, shuffle[2])}
    return {k: DevicePrefetcher(v, prefetch_device) for k, v in out.items()} if pin else out


class DevicePrefetcher:
    """A DataLoader iterated ONE BATCH AHEAD (real-data input path): while the train step of minibatch k runs, minibatch
    k+1 is collated (into pinned host memory if the loader pins) and copied to the device on a separate HIP stream; the consumer's
    stream waits on the copy's event only when it takes the batch.  The reference copies every tensor of every minibatch with a
    blocking .to(device) inside the step loop (vae_reg_GP.py:420-423).  Yields the same sample dictionaries, tensors on the device.
    On a CPU device it is a pass-through."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.dataset = loader.dataset                           # len(loader.dataset) as the train loop uses it
        self.batch_sampler = getattr(loader, 'batch_sampler', None)
        self.collate_fn = getattr(loader, 'collate_fn', None)
        self._stream = torch.cuda.Stream(self.device) if self.device.type == 'cuda' else None

    def __len__(self):
        return len(self.loader)

    def _stage(self, sample):
        if sample is None:
            return None
        with torch.cuda.stream(self._stream):
            out = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in sample.items()}
            ev = torch.cuda.Event()
            ev.record(self._stream)
        return out, ev

    def __iter__(self):
        if self._stream is None:
            yield from self.loader
            return
        it = iter(self.loader)
        nxt = self._stage(next(it, None))
        while nxt is not None:
            cur, ev = nxt
            nxt = self._stage(next(it, None))                    # the next copy is queued before this batch is handed out
            here = torch.cuda.current_stream(self.device)
            here.wait_event(ev)
            for v in cur.values():
                if torch.is_tensor(v):
                    v.record_stream(here)                        # allocated on the copy stream, used on the consumer's
            yield cur


class DeviceResidentData:
    """A whole data set kept in HBM: volumes (N, X, Y, Z) fp32, covariates (N, C) fp32, subject ids.
    Iterating yields the same sample dictionaries as the DataLoaders above, already on the device;
    `rank`/`world` give each data-parallel rank a contiguous slice of every (global) minibatch."""

    def __init__(self, volumes, covariates, subjid, batch_size, shuffle=False, seed=0, device=None, rank=0, world=1,
                 drop_last=True):
        dev = device if device is not None else volumes.device
        self.volumes = volumes.to(dev); self.covariates = covariates.to(dev); self.subjid = subjid.to(dev)
        self.batch_size, self.shuffle, self.rank, self.world, self.drop_last = batch_size, shuffle, rank, world, drop_last
        self.gen = torch.Generator().manual_seed(seed)          # identical permutation on every rank
        assert batch_size % world == 0
        self.dataset = self                                     # len(loader.dataset) as the train loop uses it

    def __len__(self):
        return self.volumes.shape[0]

    def __iter__(self):
        N = len(self)
        order = torch.randperm(N, generator=self.gen) if self.shuffle else torch.arange(N)
        local = self.batch_size // self.world
        for s in range(0, N - (self.batch_size - 1 if self.drop_last else 0), self.batch_size):
            idx = order[s:s + self.batch_size]
            idx = idx[self.rank * local:(self.rank + 1) * local].to(self.volumes.device)
            yield {'volume': self.volumes[idx], 'covariates': self.covariates[idx], 'subjid': self.subjid[idx],
                   'vol_num': idx.double()}
