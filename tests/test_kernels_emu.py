"""Host build of the HIP kernel sources (tests/emu, g++ -DVG_EMU) against PyTorch references on
tiny shapes: checks tile/halo/index arithmetic and barrier structure without a GPU.  The -m gpu
twin (tests/test_kernels_gpu.py) runs the same bodies on libvaegam_hip.so."""
import os
import subprocess

import pytest
import torch

import vae_gam_amd  # noqa: F401
from vae_gam_amd import _lib
import kernel_cases as K

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu')


@pytest.fixture(scope='module', autouse=True)
def emu_lib():
    import emu_inject
    prev = emu_inject.inject_emu()
    yield
    emu_inject.restore(prev)


@pytest.mark.parametrize('name,spec,isz', K.LAYERS, ids=[l[0] for l in K.LAYERS])
def test_layer_bn_relu(name, spec, isz):
    K.run_layer_case('cpu', name, spec, isz, with_bn=True, relu_in=(name != 'conv1'), groups=2 if spec.kind == 'convt' else 1)


@pytest.mark.parametrize('name,spec,isz', [K.LAYERS[1], K.LAYERS[3], K.LAYERS[6], K.LAYERS[8]], ids=['conv2', 'conv4', 'convt2', 'convt4'])
def test_layer_relu_only(name, spec, isz):
    K.run_layer_case('cpu', name, spec, isz, with_bn=False, relu_in=True, groups=1, seed=3)


@pytest.mark.parametrize('name,spec,isz', K.WIDE_LAYERS, ids=[l[0] for l in K.WIDE_LAYERS])
def test_layer_wide_rows(name, spec, isz):
    K.run_layer_case('cpu', name, spec, isz, with_bn=name.startswith(('conv1', 'conv3', 'convt3', 'convt5')), relu_in=not name.startswith('conv1'),
                     groups=2 if spec.kind == 'convt' else 1, seed=11)


def test_first_layer_input_is_data():
    name, spec, isz = K.LAYERS[0]
    K.run_layer_case('cpu', name, spec, isz, with_bn=True, relu_in=False, groups=1, input_is_data=True, seed=5)


def test_fused_bn_statistics_and_bias_sum():
    K.run_fused_stats_case('cpu', K.LAYERS[8][1], K.LAYERS[8][2])        # convt4-shaped (5x3x3, stride 2)
    K.run_fused_stats_case('cpu', K.LAYERS[6][1], K.LAYERS[6][2], seed=1)


def test_gam_elbo():
    K.run_gam_case('cpu', C=3, B=3, V=1500)


def test_latent_sample_kl():
    K.run_latent_case('cpu', B=5, L=32, G=4)
    K.run_latent_case('cpu', B=7, L=70, G=2, seed=3, tiny_d=True)


def test_elbo_loss():
    K.run_loss_case('cpu', B=32, C=3)
    K.run_loss_case('cpu', B=300, C=8, seed=1)


def test_linear_act_accumulates_into_grad():
    K.run_linear_case('cpu')


@pytest.mark.parametrize('case', K.FC_GEMM_CASES, ids=lambda c: '%dx%dx%d%s%s' % (c[0], c[1], c[2], '-split' if c[7] else '', '-batch' if c[6] > 1 else ''))
def test_fc_gemm_matches_float64_product(case):
    M, N, Kd, a_kc, b_kc, flags, batch, ksplit = case
    K.run_fc_gemm_case('cpu', M, N, Kd, a_kc, b_kc, flags, batch, ksplit)


def test_gam_elbo_no_covariates():
    K.run_gam_case('cpu', C=0, B=2, V=700, seed=2)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_adam(dtype):
    K.run_adam_case('cpu', dtype)


@pytest.mark.parametrize('n', [6, 32, 64])
def test_cholesky(n):
    K.run_cholesky_case('cpu', batch=3, n=n)


@pytest.mark.parametrize('B,n,jitter', [(12, 6, 0.0), (17, 6, 0.0), (7, 12, 1e-5), (130, 6, 0.0)])    # 130: the large-batch path (blocked Cholesky, slab solves)
def test_gain_block_matches_float64_oracle(B, n, jitter):
    K.run_gain_case('cpu', B=B, n=n, jitter=jitter, seed=B)


@pytest.mark.parametrize('kind,force', [
    ('convt3_fwd', (8, 2, 11, 8, 0)),      # whole planes (PHB = PH = 11), 8 waves, two channel chunks, single-buffered
    ('convt3_fwd', (8, 2, 4, 16, 1)),      # row slabs of 4 rows (3 slabs), double-buffered
    ('convt3_fwd', (4, 1, 6, 4, 0)),       # 4-wave workgroups, 2 slabs, 4 chunks
    ('convt3_bwd', (8, 1, 9, 8, 0)),       # data gradient, whole planes
    ('convt3_bwd', (4, 2, 3, 4, 1)),       # data gradient, 4 waves, slabs of 3 rows, mask + double buffer
    ('convt4_fwd', (8, 1, 10, 8, 0)),      # 4 parity classes, whole planes, statistics
    ('convt4_fwd', (8, 2, 5, 8, 1)),       # ... row slabs
    ('convt4_fwd', (4, 1, 4, 8, 0)),       # ... 4-wave workgroups
])
def test_conv_mm_pinned_tiles(kind, force):
    K.run_conv_mm_plan_case('cpu', kind, force)
