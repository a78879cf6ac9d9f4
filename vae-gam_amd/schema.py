"""Covariate schema and network geometry of the VAE-GAM (host-side description only).

The reference hard-codes eight covariates, six inducing points and one image shape
(vae_reg_GP.py:32,68,308,352,377).  This module states the same rules as data so
that the 3-, 8- and 12-covariate configurations and the 82x98x70 geometry share one code path,
and reduces exactly to the reference for num_covariates <= 8 at 41x49x35.
"""
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

from .ops import ConvSpec

REF_NAMES = ['task', 'x', 'y', 'z', 'xrot', 'yrot', 'zrot', 'sex']                      # vae_reg_GP.py:68
REF_IMG_KEYS = ['base', 'task', 'x_mot', 'y_mot', 'z_mot', 'pitch_mot', 'roll_mot', 'yaw_mot', 'sex',
                'full_rec']                                                              # vae_reg_GP.py:308-309
REF_CSV_COLS = ['x', 'y', 'z', 'rot_x', 'rot_y', 'rot_z']                                # utils.py:50


@dataclass(frozen=True)
class Covariate:
    name: str
    gp: bool       # continuous: linear gain + sparse-GP gain (vae_reg_GP.py:352)
    hrf: bool      # gain convolved with the HRF along the batch axis (vae_reg_GP.py:377)
    img_key: str   # key in the dict returned with return_latent_rec=True


def covariate_schema(num_covariates: int, neural_covariates: bool = True) -> List[Covariate]:
    """Roles the reference assigns by position: GP iff 1 < i < 8, HRF iff neural and i < C-6.
    Beyond eight covariates (no reference counterpart) extra continuous covariates c1.. follow
    `task`, every covariate but the first and last gets a GP term, the HRF rule stays positional."""
    C = int(num_covariates)
    if C < 1:
        raise ValueError('num_covariates must be >= 1')
    if C <= 8:
        names, keys = REF_NAMES[:C], REF_IMG_KEYS[1:C + 1]
        gp = [1 < i < 8 for i in range(1, C + 1)]
    else:
        extra = ['c%d' % k for k in range(1, C - 8 + 1)]
        names = ['task'] + extra + REF_NAMES[1:]
        keys = ['task'] + extra + REF_IMG_KEYS[2:9]
        gp = [1 < i < C for i in range(1, C + 1)]
    hrf = [bool(neural_covariates) and i < (C - 6) for i in range(1, C + 1)]
    return [Covariate(n, g, h, k) for n, g, h, k in zip(names, gp, hrf, keys)]


def parameter_sets(num_covariates: int, neural_covariates: bool = True) -> List[Covariate]:
    """Gain-parameter sets that are INSTANTIATED: the reference always creates all eight
    (vae_reg_GP.py:68-172) even when fewer covariates are used."""
    return covariate_schema(max(8, num_covariates), neural_covariates)


@dataclass(frozen=True)
class NetGeometry:
    img: Tuple[int, int, int]
    nf: int
    enc: Tuple[ConvSpec, ...]          # conv1..conv5
    dec: Tuple[ConvSpec, ...]          # convt1..convt5
    dec_seed: Tuple[int, int, int]     # spatial size fc8's output is viewed as

    def enc_sizes(self):
        s = [self.img]
        for sp in self.enc:
            s.append(sp.out_size(s[-1]))
        return s

    def dec_sizes(self):
        s = [self.dec_seed]
        for sp in self.dec:
            s.append(sp.out_size(s[-1]))
        return s

    @property
    def enc_flat(self):
        return 2 * self.nf * int(np.prod(self.enc_sizes()[-1]))

    @property
    def dec_flat(self):
        return 2 * self.nf * int(np.prod(self.dec_seed))


def net_geometry(img: Sequence[int], nf: int = 8) -> NetGeometry:
    img = tuple(int(v) for v in img)
    k3 = (3, 3, 3)
    enc = (ConvSpec('conv', 1, nf, k3, 1), ConvSpec('conv', nf, nf, k3, 2), ConvSpec('conv', nf, 2 * nf, k3, 1),
           ConvSpec('conv', 2 * nf, 2 * nf, k3, 2), ConvSpec('conv', 2 * nf, 2 * nf, k3, 1))      # :189-193
    if img == (41, 49, 35):
        dec = (ConvSpec('convt', 2 * nf, 2 * nf, k3, 1),
               ConvSpec('convt', 2 * nf, 2 * nf, k3, 2, (1, 0, 1), (1, 0, 1)),
               ConvSpec('convt', 2 * nf, nf, k3, 1),
               ConvSpec('convt', nf, nf, (5, 3, 3), 2),
               ConvSpec('convt', nf, 1, k3, 1))                                                   # :211-215
        seed = (6, 8, 5)                                                                          # :259
    elif img == (82, 98, 70):
        # 16x20x13 -> 18x22x15 -> 37x45x31 -> 39x47x33 -> (k4,s2) 80x96x68 -> 82x98x70
        dec = (ConvSpec('convt', 2 * nf, 2 * nf, k3, 1), ConvSpec('convt', 2 * nf, 2 * nf, k3, 2),
               ConvSpec('convt', 2 * nf, nf, k3, 1), ConvSpec('convt', nf, nf, (4, 4, 4), 2),
               ConvSpec('convt', nf, 1, k3, 1))
        seed = (16, 20, 13)
    elif img == (21, 21, 21):
        # smallest volume the encoder admits; used by the CPU tests (host build of the kernels) only
        dec = (ConvSpec('convt', 2 * nf, 2 * nf, k3, 1), ConvSpec('convt', 2 * nf, 2 * nf, k3, 2),
               ConvSpec('convt', 2 * nf, nf, k3, 1), ConvSpec('convt', nf, nf, k3, 2), ConvSpec('convt', nf, 1, k3, 1))
        seed = (1, 1, 1)
    else:
        raise ValueError('no network geometry defined for image shape %r' % (img,))
    from dataclasses import replace
    enc = tuple(replace(sp, name='conv%d' % (i + 1)) for i, sp in enumerate(enc))
    dec = tuple(replace(sp, name='convt%d' % (i + 1)) for i, sp in enumerate(dec))
    g = NetGeometry(img, nf, enc, dec, seed)
    assert g.dec_sizes()[-1] == img, (g.dec_sizes(), img)
    return g
