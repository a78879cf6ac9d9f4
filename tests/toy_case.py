"""Shared toy case (21x21x21 volumes) for the CPU tests that drive the PRODUCT model through the host build
of the HIP kernels (tests/emu) -- index arithmetic, autograd glue and data-parallel bookkeeping without a GPU."""
import os
import subprocess

import numpy as np
import torch

import vae_gam_amd  # noqa: F401
from vae_gam_amd import _lib
from vae_gam_amd.vae_reg_GP import VAE

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu')
IMG = (21, 21, 21)


def load_emu_library():
    import emu_inject
    emu_inject.inject_emu()


def make_inputs(B, C, seed=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    V = int(np.prod(IMG))
    x = np.clip(0.5 + 0.25 * rng.normal(size=(B,) + IMG), 0, 1).astype(np.float32)
    cont = rng.normal(size=(B, 6)); cont[0] = 6.0; cont[1] = -4.0
    task = (np.arange(B) % 2).astype(np.float64); sex = (np.arange(B) >= B // 2).astype(np.float64)
    cov = np.stack([task, *cont.T, sex], 1)[:, :C].astype(np.float32)
    xu = [[float(cont[:, j].min()) - 1e-3, float(cont[:, j].max()) + 1e-3] for j in range(6)]
    glm = rng.uniform(size=(V, 8)); glm = glm / glm.max(0, keepdims=True)
    glm = np.concatenate([np.arange(V, dtype=np.float64)[:, None], glm], 1)
    return torch.from_numpy(x), torch.from_numpy(cov), xu, glm


def make_model(C, xu, glm, dp=None, seed=1, dp_gain='global'):
    torch.manual_seed(seed)
    return VAE(num_covariates=C, glm_maps=glm, xu_ranges=xu, device_name='cpu', img_shape=IMG, data_parallel=dp, dp_gain=dp_gain)
