"""Host-side surface of the drop-in (no GPU): covariate schema, seeded construction, checkpoint dictionary,
optimizer state format, CLI flags, data classes, synthetic generator, HRF kernel."""
import json
import os
import sys

import numpy as np
import pytest
import torch

import vae_gam_amd  # noqa: F401
from vae_gam_amd import DataClass_GP, multsubj_reg_run_GP, schema, synthetic, utils
from vae_gam_amd.vae_reg_GP import VAE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def small_ds():
    return synthetic.make_dataset(num_subjects=2, vols_per_subject=6, num_covariates=8, seed=1)


def test_schema_reduces_to_reference_rules():
    s8 = schema.covariate_schema(8)
    assert [c.name for c in s8] == ['task', 'x', 'y', 'z', 'xrot', 'yrot', 'zrot', 'sex']
    assert [c.gp for c in s8] == [False, True, True, True, True, True, True, False]          # 1 < i < 8
    assert [c.hrf for c in s8] == [True] + [False] * 7                                        # i < C-6
    assert [c.hrf for c in schema.covariate_schema(8, neural_covariates=False)] == [False] * 8
    s3 = schema.covariate_schema(3)
    assert [c.name for c in s3] == ['task', 'x', 'y'] and not any(c.hrf for c in s3)
    s12 = schema.covariate_schema(12)
    assert len(s12) == 12 and s12[0].name == 'task' and s12[-1].name == 'sex' and sum(c.gp for c in s12) == 10
    g = schema.net_geometry((41, 49, 35))
    assert g.enc_flat == 3072 and g.dec_flat == 3840 and g.dec_sizes()[-1] == (41, 49, 35)   # vae_reg_GP.py:197,210
    assert schema.net_geometry((82, 98, 70)).dec_sizes()[-1] == (82, 98, 70)


def test_hrf_kernel_values():
    h = utils.hrf(np.arange(0, 20, 1.4))
    ref = [0, .040384, .318517, .593309, .6, .408907, .180511, .010367, -.079018, -.104684, -.094269, -.070497,
           -.046685, -.028236, -.015885]
    np.testing.assert_allclose(h, ref, atol=1e-6)


def test_seeded_model_matches_reference_parameter_order_and_checkpoint(small_ds, tmp_path):
    meta = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'ref_B4_C8.json')))
    torch.manual_seed(1)
    m = VAE(num_covariates=8, glm_maps=small_ds['glm'], xu_ranges=small_ds['xu_ranges'], device_name='cpu', save_dir=str(tmp_path))
    assert [n for n, _ in m.named_parameters()] == meta['param_order']
    assert sum(p.numel() for p in m.parameters()) == 1564424
    assert m.epsilon.dtype == torch.float64 and m.glm_maps.dtype == torch.float64
    m.epoch = 7
    m.save_state('ck.tar')
    ck = torch.load(os.path.join(tmp_path, 'ck.tar'), weights_only=False)
    ref = meta['checkpoint_keys']
    assert set(ck.keys()) == set(ref.keys())
    for layer in ('conv1', 'convt5', 'bn1', 'fc8'):
        assert sorted(ck[layer].keys()) == ref[layer]
    assert {k: sorted(v.keys()) for k, v in ck['gp_params'].items()} == ref['gp_params']
    assert sorted(ck['optimizer_state']['param_groups'][0].keys()) == ref['optimizer_state']['param_groups_keys']
    assert len(ck['optimizer_state']['param_groups'][0]['params']) == 97
    # round trip into a differently seeded model: values are copied INTO the live parameters
    torch.manual_seed(5)
    m2 = VAE(num_covariates=8, glm_maps=small_ds['glm'], xu_ranges=small_ds['xu_ranges'], device_name='cpu', save_dir=str(tmp_path))
    ids_before = {n: p.data_ptr() for n, p in m2.named_parameters()}
    m2.load_state(os.path.join(tmp_path, 'ck.tar'))
    assert m2.epoch == 7
    for (n, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.equal(p, q), n
        assert q.data_ptr() == ids_before[n]                 # still the tensors the optimiser updates
    assert m2.gp_params['x']['qu_S'] is m2.qu_S_x


def test_gp_jitter_travels_with_the_checkpoint_and_reaches_the_posterior_export(small_ds, tmp_path):
    """gp_jitter (the Ku + jitter I remedy for dense inducing grids) is part of the model a checkpoint describes: saved as an extra key
    when set, restored by load_state (which also drops the constants / graphs derived from the replaced state), exposed as --gp_jitter,
    and used by gp.posterior_diag_batched -- the posterior plot_GPs exports is the one the training kernel evaluated."""
    from vae_gam_amd import gp
    torch.manual_seed(1)
    m = VAE(num_covariates=8, glm_maps=small_ds['glm'], xu_ranges=small_ds['xu_ranges'], device_name='cpu', save_dir=str(tmp_path),
            num_inducing_pts=64, gp_jitter=1e-4)
    m.save_state('ckj.tar')
    ck = torch.load(os.path.join(tmp_path, 'ckj.tar'), weights_only=False)
    assert ck['gp_jitter'] == 1e-4
    m2 = VAE(num_covariates=8, glm_maps=small_ds['glm'], xu_ranges=small_ds['xu_ranges'], device_name='cpu', save_dir=str(tmp_path),
             num_inducing_pts=64)
    m2._gain_const_cache['stale'] = object(); m2._graphs['stale'] = object()
    m2.load_state(os.path.join(tmp_path, 'ckj.tar'))
    assert m2.gp_jitter == 1e-4 and not m2._gain_const_cache and not m2._graphs
    assert multsubj_reg_run_GP.build_parser().parse_args(['--gp_jitter', '1e-4']).gp_jitter == 1e-4
    # 64 points on [-4, 6]: spacing 0.16 against a length scale of ~2 -- the plain inverse is garbage, the jittered posterior is not
    xu = m.gp_params['x']['xu'].double().unsqueeze(0)
    kv = torch.tensor([1.1], dtype=torch.float64); ls = torch.tensor([2.0], dtype=torch.float64)
    qm = m.gp_params['x']['qu_m'].detach().double(); qS = m.gp_params['x']['qu_S'].detach().double().unsqueeze(0)
    xq = torch.linspace(-3.5, 5.5, 50, dtype=torch.float64).unsqueeze(0)
    fj, vj = gp.posterior_diag_batched(xu, kv, ls, qm, qS, xq, jitter=1e-4)
    fb, Sg = gp.posterior_batched(xu, kv, ls, qm, qS, xq, jitter=1e-4)
    np.testing.assert_allclose(fj.numpy(), fb.numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(vj.numpy(), Sg.diagonal(dim1=1, dim2=2).numpy(), rtol=1e-8, atol=1e-8)
    # the same quantities from the definition, float64: A = Knu^T (Ku1 + jI)^-1, var = k_var + diag(A (S - k_var (Ku1 + jI)) A^T)
    d = (xu[0].unsqueeze(1) - xq[0].unsqueeze(0))
    k1 = lambda dist: torch.exp(-(dist / (np.sqrt(2) * 2.0)) ** 2)
    ku = k1(xu[0].unsqueeze(0) - xu[0].unsqueeze(1)) + 1e-4 * torch.eye(64, dtype=torch.float64)
    A = torch.linalg.solve(ku, k1(d)).T
    # (the implementation rebuilds the grid as xu[0] + k*step like gp.py:92-94, this check uses the fp32-rounded grid points themselves:
    #  1e-7 apart, times cond(Ku1 + jI) ~ 1e5)
    np.testing.assert_allclose(fj[0].numpy(), (A @ qm[0]).numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(vj[0].numpy(), (1.1 + ((A @ (qS[0] - 1.1 * ku)) * A).sum(1)).numpy(), rtol=1e-4, atol=1e-4)
    f0, _ = gp.posterior_diag_batched(xu, kv, ls, qm, qS, xq)            # the plain inverse on this grid: not a posterior
    assert not np.allclose(f0.numpy(), fj.numpy(), rtol=1e-2, atol=1e-2)


def test_checkpoint_loads_into_torch_adam(small_ds, tmp_path):
    """optimizer_state is torch.optim.Adam's own format: the reference's load_state can consume it."""
    torch.manual_seed(1)
    m = VAE(num_covariates=8, glm_maps=small_ds['glm'], xu_ranges=small_ds['xu_ranges'], device_name='cpu', save_dir=str(tmp_path))
    m.optimizer.step_count = 3
    g = m.optimizer.groups[torch.float32]; g['m'].normal_(); g['v'].uniform_()
    sd = m.optimizer.state_dict()
    ref_opt = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in m.parameters()], lr=1e-3)
    ref_opt.load_state_dict(sd)
    st = ref_opt.state_dict()['state']
    assert len(st) == 97 and float(st[0]['step']) == 3.0
    m.optimizer.load_state_dict(ref_opt.state_dict())
    assert m.optimizer.step_count == 3


def test_forward_without_gpu_raises(small_ds):
    torch.manual_seed(1)
    m = VAE(num_covariates=8, glm_maps=small_ds['glm'], xu_ranges=small_ds['xu_ranges'], device_name='cpu')
    x = torch.from_numpy(small_ds['volumes'][:2]); cov = torch.from_numpy(small_ds['covariates'][:2])
    with pytest.raises(RuntimeError, match='no CPU path'):
        m.forward(torch.zeros(2, dtype=torch.int64), cov, x, 'train', train_mode=False)


def test_cli_flags_match_reference():
    a = multsubj_reg_run_GP.build_parser().parse_args([])
    assert (a.batch_size, a.epochs, a.seed, a.save_freq, a.test_freq, a.split) == (32, 300, 1, 100, 200, 98)
    assert (a.glm_reg_scale, a.num_inducing_pts, a.gp_kl_scale) == (1.0, 6, 10.0)
    assert a.from_ckpt is False and a.recons_only is False and a.neural_covariates is True
    b = multsubj_reg_run_GP.build_parser().parse_args(['--from_ckpt', '--neural_covariates', 'no', '--batch-size', '64'])
    assert b.from_ckpt is True and b.neural_covariates is False and b.batch_size == 64


def test_dataset_sample_dictionary_and_loaders(small_ds, tmp_path):
    csv, glm_csv = synthetic.write_csvs(small_ds, str(tmp_path))
    ds = DataClass_GP.FMRIDataset(csv, transform=DataClass_GP.ToTensor())
    assert len(ds) == 12
    s = ds[7]
    assert set(s.keys()) == {'covariates', 'volume', 'subjid', 'vol_num'}
    assert s['covariates'].shape == (8,) and s['covariates'].dtype == torch.float32
    assert s['volume'].shape == (41, 49, 35) and s['volume'].dtype == torch.float32
    assert s['subjid'].dtype == torch.int64 and int(s['subjid']) == 1 and s['vol_num'].dtype == torch.float64
    np.testing.assert_allclose(s['volume'].numpy(), small_ds['volumes'][7], atol=1e-6)       # /3284.5 undone exactly
    np.testing.assert_allclose(s['covariates'].numpy(), small_ds['covariates'][7], atol=1e-6)
    loaders = DataClass_GP.setup_data_loaders(batch_size=4, train_csv=csv, test_csv=csv)
    assert set(loaders) == {'Shuffled_train', 'UnShuffled_train', 'test'}
    b = next(iter(loaders['UnShuffled_train']))
    assert b['volume'].shape == (4, 41, 49, 35) and b['covariates'].shape == (4, 8)
    assert utils.get_xu_ranges([csv, csv])[0][0] == pytest.approx(small_ds['xu_ranges'][0][0], abs=1e-5)
    glm = np.loadtxt(glm_csv, delimiter=',', skiprows=1)
    assert glm.shape == (70315, 9)


def test_nifti1_reader(tmp_path):
    import struct
    a = np.arange(2 * 3 * 4 * 5, dtype=np.int16).reshape((2, 3, 4, 5), order='F')
    hdr = bytearray(352)
    struct.pack_into('<i', hdr, 0, 348)
    struct.pack_into('<8h', hdr, 40, 4, 2, 3, 4, 5, 1, 1, 1)
    struct.pack_into('<2h', hdr, 70, 4, 16)
    struct.pack_into('<f', hdr, 108, 352.0)
    struct.pack_into('<2f', hdr, 112, 2.0, 1.0)
    path = str(tmp_path / 't.nii')
    open(path, 'wb').write(bytes(hdr) + a.tobytes(order='F'))
    out = DataClass_GP.read_nifti1(path)
    np.testing.assert_allclose(out, a * 2.0 + 1.0)


def test_synthetic_set_is_seeded_and_has_the_control_signal():
    a = synthetic.make_dataset(num_subjects=1, vols_per_subject=30, num_covariates=3, seed=4)
    b = synthetic.make_dataset(num_subjects=1, vols_per_subject=30, num_covariates=3, seed=4)
    assert np.array_equal(a['volumes'], b['volumes']) and np.array_equal(a['covariates'], b['covariates'])
    assert a['volumes'].min() >= 0 and a['volumes'].max() <= 1
    task = a['covariates'][:, 0]
    assert task[0] == 1 and task[14] == 0                       # ON first, 20 s blocks at TR 1.4 (utils.py:93-111)
    sig = synthetic.large3_signal()
    inside = np.zeros_like(sig, dtype=bool); inside[15:25, 34:47, 9:22] = True
    assert sig[inside].sum() > 0 and not sig[~inside].any()
    on = a['volumes'][task == 1].mean(0); off = a['volumes'][task == 0].mean(0)
    assert (on - off)[sig > 0].mean() > 0.2                      # the glyph is there when the block is ON
    assert a['covariates'][:, 1].min() < -3 and a['covariates'][:, 1].max() > 3   # wide range keeps Ku conditioned (H2)


def test_nifti1_writer_round_trip_and_reference_geometry(tmp_path):
    import struct
    from vae_gam_amd import nifti
    rng = np.random.default_rng(0)
    a = rng.standard_normal((5, 7, 3)).astype(np.float32)
    p = str(tmp_path / 'm.nii')
    nifti.write_nifti1(p, a)
    np.testing.assert_array_equal(DataClass_GP.read_nifti1(p), a)
    pz = str(tmp_path / 'm.nii.gz')
    nifti.write_nifti1(pz, a.astype(np.float64))
    np.testing.assert_array_equal(DataClass_GP.read_nifti1(pz), a)
    # geometry of a reference file (pixdim, sform) survives; its int16 datatype / scaling do not leak into the float map
    ref = bytearray(352)
    struct.pack_into('<i', ref, 0, 348)
    struct.pack_into('<8h', ref, 40, 3, 5, 7, 3, 1, 1, 1, 1)
    struct.pack_into('<2h', ref, 70, 4, 16)
    struct.pack_into('<8f', ref, 76, 1.0, 3.0, 3.0, 3.5, 1.4, 1.0, 1.0, 1.0)
    struct.pack_into('<f', ref, 108, 352.0)
    struct.pack_into('<2f', ref, 112, 2.0, 1.0)
    struct.pack_into('<h', ref, 254, 2)
    struct.pack_into('<4f', ref, 280, -3.0, 0.0, 0.0, 90.0)
    refp = str(tmp_path / 'ref.nii')
    open(refp, 'wb').write(bytes(ref) + np.zeros(5 * 7 * 3, np.int16).tobytes())
    p2 = str(tmp_path / 'withref.nii')
    nifti.write_nifti1(p2, a, reference=refp)
    np.testing.assert_array_equal(DataClass_GP.read_nifti1(p2), a)
    raw, en = nifti.read_header(p2)
    assert struct.unpack(en + '8f', raw[76:108])[1:4] == (3.0, 3.0, 3.5)
    assert struct.unpack(en + '4f', raw[280:296]) == (-3.0, 0.0, 0.0, 90.0)
    assert struct.unpack(en + '2h', raw[70:74]) == (16, 32)


def test_nifti1_reader_and_writer_against_spec_fixtures(tmp_path):
    """the NIfTI-1 reader / writer against files built field by field from the format specification
    (oracle/gen_nifti_fixture.py), not against each other: decode of both byte orders, and a byte-identical encode."""
    from vae_gam_amd import nifti
    GOLDEN = os.path.join(ROOT, 'tests', 'golden')
    want = np.fromfunction(lambda i, j, k: 100 * i + 10 * j + k, (3, 4, 5)).astype(np.float32)
    for tag in ('le', 'be'):
        got = DataClass_GP.read_nifti1(os.path.join(GOLDEN, 'nifti1_3x4x5_f32_%s.nii' % tag))
        assert got.shape == (3, 4, 5)
        np.testing.assert_array_equal(np.asarray(got, np.float32), want)
    p = str(tmp_path / 'w.nii')
    nifti.write_nifti1(p, want)
    assert open(p, 'rb').read() == open(os.path.join(GOLDEN, 'nifti1_3x4x5_f32_le.nii'), 'rb').read()
    # with a reference file the geometry comes from it, in its byte order
    p2 = str(tmp_path / 'w_be.nii')
    nifti.write_nifti1(p2, want, reference=os.path.join(GOLDEN, 'nifti1_3x4x5_f32_be.nii'))
    assert open(p2, 'rb').read() == open(os.path.join(GOLDEN, 'nifti1_3x4x5_f32_be.nii'), 'rb').read()


def test_device_prefetcher_is_a_pass_through_on_cpu(small_ds, tmp_path):
    """DataClass_GP.DevicePrefetcher on a CPU device hands out the wrapped loader's batches unchanged; setup_data_loaders without
    prefetch_device returns plain DataLoaders (the reference's)."""
    csv, _ = synthetic.write_csvs(small_ds, str(tmp_path))
    loaders = DataClass_GP.setup_data_loaders(batch_size=4, train_csv=csv, test_csv=csv)
    assert type(loaders['test']).__name__ == 'DataLoader'
    pf = DataClass_GP.DevicePrefetcher(loaders['UnShuffled_train'], 'cpu')
    assert len(pf) == len(loaders['UnShuffled_train']) and pf.dataset is loaders['UnShuffled_train'].dataset
    for a, b in zip(pf, loaders['UnShuffled_train']):
        assert sorted(a) == sorted(b)
        for k in a:
            assert torch.equal(a[k], b[k])


def test_mk_avg_maps_from_device_sums(tmp_path):
    """Subject means and the grand mean (mean of subject means, build_model_recons.py:86-99) from the sums reconstruct() leaves."""
    import pandas as pd
    from vae_gam_amd import build_model_recons as R

    class FakeModel:
        epoch = 7
        img_shape = (3, 4, 2)
    V = 24
    rng = np.random.default_rng(1)
    per_vol = {0: rng.standard_normal((5, V)), 1: rng.standard_normal((3, V))}            # subject -> (volumes, V)
    m = FakeModel()
    sums = {'base': torch.tensor(np.stack([per_vol[0].sum(0), per_vol[1].sum(0)])),
            'x_mot': torch.zeros(2, V, dtype=torch.float64), 'full_rec': torch.ones(2, V, dtype=torch.float64)}
    m.recon_sums = (sums, torch.tensor([5.0, 3.0], dtype=torch.float64))
    csv = str(tmp_path / 'd.csv')
    pd.DataFrame({'i': range(8), 'subjid': ['sA'] * 5 + ['sB'] * 3, 'vol': list(range(5)) + list(range(3)),
                  'nii_path': ['a.npy'] * 5 + ['b.npy'] * 3}).to_csv(csv, index=False)
    out = R.mk_avg_maps(csv, m, str(tmp_path))
    assert set(out) == {'base', 'full_rec'}                                               # motion maps only on request
    want = 0.5 * (per_vol[0].mean(0) + per_vol[1].mean(0))
    np.testing.assert_allclose(out['base'].reshape(-1), want, rtol=1e-6, atol=1e-6)
    d = tmp_path / 'reconstructions' / '007_avg_model_recons'
    np.testing.assert_allclose(DataClass_GP.read_nifti1(str(d / 'sB' / 'base_avg.nii')).reshape(-1), per_vol[1].mean(0), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(DataClass_GP.read_nifti1(str(d / 'base_avg.nii')).reshape(-1), want, rtol=1e-5, atol=1e-6)
    assert 'x_mot' in R.mk_avg_maps(csv, m, str(tmp_path), mk_motion_maps=True)


def test_gp_posterior_diagonal_equals_full_posterior_diagonal():
    from vae_gam_amd import gp
    g = torch.Generator().manual_seed(0)
    K, n, N = 3, 6, 40
    xu = torch.stack([torch.linspace(-4.0 - k, 6.0 + k, n) for k in range(K)]).double()
    kvar = (torch.rand(K, generator=g) + 0.2).double(); ls = (torch.rand(K, generator=g) * 2 + 0.8).double()
    qu_m = torch.randn(K, n, generator=g).double()
    r = torch.randn(K, n, n, generator=g).double(); qu_S = r @ r.transpose(1, 2) + 0.5 * torch.eye(n).double()
    xq = (torch.rand(K, N, generator=g) * 9 - 4).double()
    f_full, S_full = gp.posterior_batched(xu, kvar, ls, qu_m, qu_S, xq)
    f_d, v_d = gp.posterior_diag_batched(xu, kvar, ls, qu_m, qu_S, xq)
    np.testing.assert_allclose(f_d.numpy(), f_full.numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(v_d.numpy(), torch.diagonal(S_full, dim1=1, dim2=2).numpy(), rtol=1e-9, atol=1e-10)


def test_plot_GPs_csv_export_matches_per_covariate_posterior(small_ds, tmp_path):
    """SURVEY 8f-2 (vae_reg_GP.py:641-673): sorted xq / mean / vars per continuous covariate == the reference's
    recipe evaluated with the full (N x N) posterior of gp.GP."""
    import pandas as pd
    from vae_gam_amd import gp
    csv, _ = synthetic.write_csvs(small_ds, str(tmp_path))
    torch.manual_seed(2)
    m = VAE(num_covariates=8, glm_maps=small_ds['glm'], xu_ranges=small_ds['xu_ranges'], device_name='cpu', save_dir=str(tmp_path))
    m.epoch = 4
    out = m.plot_GPs(csv_file=csv, save_dir=str(tmp_path))
    assert sorted(out) == sorted(['x', 'y', 'z', 'xrot', 'yrot', 'zrot'])
    data = pd.read_csv(csv)
    name, col = 'yrot', 'rot_y'
    f = pd.read_csv(str(tmp_path / '004_GP_plots' / ('004_GP_%s_full.csv' % name)), index_col=0)
    assert list(f.columns) == ['xq', 'mean', 'vars'] and np.all(np.diff(f['xq'].to_numpy()) >= 0)
    P = m.gp_params[name]
    xq = torch.from_numpy(data[col].to_numpy(dtype=np.float32)).double()
    G = gp.GP(P['xu'].double(), P['logkvar'].double().exp() + 0.1, 3.0 * torch.sigmoid(P['log_ls'].double().exp() + 0.5),
              P['qu_m'].double(), P['qu_S'].double())
    with torch.no_grad():
        f_bar, Sigma = G.evaluate_posterior(xq)
        mean = P['sa'][0].double() * xq + f_bar
        var = P['logstd'][0].double().exp() ** 2 * xq ** 2 + torch.diagonal(Sigma)
    order = f.index.to_numpy()
    np.testing.assert_allclose(f['mean'].to_numpy(), mean.numpy()[order], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(f['vars'].to_numpy(), var.numpy()[order], rtol=1e-6, atol=1e-8)


def test_glm_beta_maps_least_squares_and_scaling():
    rng = np.random.default_rng(0)
    T, R, V = 60, 7, 50
    G = rng.standard_normal((T, R))
    B = rng.standard_normal((R, V)) * 2 + 0.5
    Y = (G @ B).T + 1e-9 * rng.standard_normal((V, T))
    maps = utils.glm_beta_maps(G, Y, sex_map=np.linspace(0.1, 1.0, V))
    assert maps.shape == (R + 1, V)
    want = np.concatenate([B, np.linspace(0.1, 1.0, V)[None]], 0)
    want = want / want.max(1, keepdims=True)                        # utils.scale_beta_maps: divide each map by its maximum
    np.testing.assert_allclose(maps, want, rtol=1e-6, atol=1e-7)
    with pytest.raises(ValueError):
        utils.glm_beta_maps(G[:-1], Y)


def test_jsonl_scalar_log(small_ds, tmp_path):
    """without tensorboard the per-epoch scalars land in <save_dir>/run/<date>/scalars.jsonl; no save_dir, no files."""
    import json, glob
    m = VAE(num_covariates=8, glm_maps=small_ds['glm'], xu_ranges=small_ds['xu_ranges'], device_name='cpu', save_dir=str(tmp_path))
    m.writer.add_scalar('Loss/Train', torch.tensor(3.5), 2)
    m._log_gain_scalars()
    m.writer.close()
    files = glob.glob(str(tmp_path / 'run' / '*' / 'scalars.jsonl'))
    assert len(files) == 1
    recs = [json.loads(l) for l in open(files[0])]
    assert recs[0] == {'tag': 'Loss/Train', 'step': 2, 'value': 3.5}
    tags = {r['tag'] for r in recs}
    assert {'gain/task/sa', 'gp/x/k_var', 'gp/zrot/ls', 'epsilon/mean'} <= tags and 'gp/task/ls' not in tags
    cwd_before = set(os.listdir('.'))
    VAE(num_covariates=3, glm_maps=small_ds['glm'], xu_ranges=small_ds['xu_ranges'], device_name='cpu')
    assert set(os.listdir('.')) == cwd_before


def test_sharded_batch_sampler_ranks_take_disjoint_slices_of_the_single_process_batches():
    """dp.ShardedBatchSampler (the CLI's data-parallel loaders): per epoch every rank holds the contiguous slice of each global
    minibatch, slices are disjoint, their union in rank order IS the minibatch one process would draw, all ranks take the same
    number of equal-sized steps, and the order changes from epoch to epoch but is identical across ranks."""
    from vae_gam_amd.dp import ShardedBatchSampler
    n, gb, seed = 23, 8, 5
    one = ShardedBatchSampler(n, gb, 0, 1, True, seed)
    for world in (2, 4):
        ranks = [ShardedBatchSampler(n, gb, r, world, True, seed) for r in range(world)]
        one.set_epoch(0)
        for epoch in range(3):
            ref = list(iter(one))
            per_rank = [list(iter(s_)) for s_ in ranks]
            assert len({len(b) for b in per_rank}) == 1 and len(per_rank[0]) == len(ranks[0])
            for k, batch in enumerate(zip(*per_rank)):
                sizes = {len(x) for x in batch}
                assert len(sizes) == 1                                   # equal-sized batches: the collectives stay in step
                union = [i for part in batch for i in part]
                assert len(set(union)) == len(union)                     # disjoint
                assert union == ref[k][:len(union)]                      # rank order == the single-process minibatch
                assert len(ref[k]) - len(union) < world                  # only the short last batch loses < world samples
        assert list(iter(ShardedBatchSampler(n, gb, 0, 2, True, seed))) != list(iter(ShardedBatchSampler(n, gb, 0, 2, True, seed + 1)))
    a = ShardedBatchSampler(n, gb, 1, 2, True, seed)
    e0, e1 = list(iter(a)), list(iter(a))
    assert e0 != e1                                                      # a new permutation every epoch
    a.set_epoch(0)
    assert list(iter(a)) == e0
    u = ShardedBatchSampler(10, 4, 1, 2, False)
    assert list(iter(u)) == [[2, 3], [6, 7], [9]]                        # unshuffled: identity order, same slicing


def test_shard_loaders_rebuilds_every_loader_on_its_dataset(small_ds, tmp_path):
    import types
    from vae_gam_amd import DataClass_GP, dp as dpmod, synthetic
    csv, _ = synthetic.write_csvs(small_ds, str(tmp_path))
    loaders = DataClass_GP.setup_data_loaders(batch_size=2, train_csv=csv, test_csv=csv)
    ctx = types.SimpleNamespace(rank=1, world_size=2)
    out = dpmod.DataParallelContext.shard_loaders(ctx, loaders, 4, 1)
    assert set(out) == {'Shuffled_train', 'UnShuffled_train', 'test'}
    for k in out:
        assert out[k].dataset is loaders[k].dataset
    b = next(iter(out['UnShuffled_train']))
    assert b['volume'].shape[0] == 2 and [int(v) for v in b['vol_num']] == [2, 3]      # rows 2,3 of the first global batch of 4


def test_real_data_input_path_matches_the_reference_scripts(tmp_path):
    """SURVEY 8f-3 against the reference's own arithmetic: tests/golden/preproc_ref.npz holds what `get_beta_map_regularizer.py`
    (:73-107: FSL design matrices -> OLS beta maps -> max-scaled CSV) and `pre_proc_vaefmri.py` (:97-129: per-volume CSV, z-scored
    motion) WROTE when oracle/gen_preproc_golden.py ran them in the build container on a seeded synthetic fmriprep / FSL tree.  The
    same recipe rebuilds the inputs here; utils.read_design_mat / glm_design_columns / glm_beta_maps / preproc_table must reproduce
    the files: values to 1e-10, column names, index column and row order exactly."""
    import pandas as pd
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import gen_preproc_golden as P                                        # the recipe and the design.mat writer only
    from vae_gam_amd import utils
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'preproc_ref.npz'))
    rc = P.recipe(int(g['seed']))
    order = list(dict.fromkeys(str(s) for s in g['csv.subjid']))         # the directory order the scripts happened to walk
    assert sorted(order) == sorted(P.SUBJECTS)
    # ---- GLM maps
    designs, datas = [], []
    for s in order:
        path = str(tmp_path / (s + '_design.mat'))
        P.write_design_mat(path, rc[s]['design'])
        m = utils.read_design_mat(path)
        np.testing.assert_allclose(m, rc[s]['design'], rtol=0, atol=5e-7)        # %e text round trip
        designs.append(utils.glm_design_columns(m))
        datas.append(rc[s]['data'].reshape(-1, P.DIMS[3]))
    maps = utils.glm_beta_maps(np.concatenate(designs, 0), np.concatenate(datas, 1), sex_map=rc['sex_map'])
    assert list(g['glm.columns'][1:]) == ['task', 'x', 'y', 'z', 'xrot', 'yrot', 'zrot', 'sex']
    np.testing.assert_array_equal(g['glm.index'], np.arange(maps.shape[1]))
    np.testing.assert_allclose(maps.T, g['glm.values'], rtol=1e-10, atol=1e-12)
    glm_csv = str(tmp_path / 'glm.csv')
    pd.DataFrame(maps.T, columns=['task', 'x', 'y', 'z', 'xrot', 'yrot', 'zrot', 'sex']).to_csv(glm_csv)
    assert list(pd.read_csv(glm_csv).columns) == list(g['glm.columns'])
    # ---- per-volume CSV
    df = utils.preproc_table([dict(subjid=s, nii_path=s + '_preproc_bold_brainmasked_resampled.nii.gz', motion=rc[s]['motion'], sex=rc[s]['sex'])
                              for s in order], control=True)
    csv = str(tmp_path / 'pre.csv')
    df.to_csv(csv)
    back = pd.read_csv(csv)
    assert list(back.columns) == list(g['csv.columns'])
    np.testing.assert_array_equal(back.iloc[:, 0].to_numpy(), g['csv.index'])
    assert list(back['subjid']) == [str(s) for s in g['csv.subjid']]
    np.testing.assert_array_equal(back['volume #'].to_numpy(), g['csv.volume'])
    assert list(back['nii_path']) == [str(s) for s in g['csv.nii_name']]
    np.testing.assert_allclose(back[['task', 'x', 'y', 'z', 'rot_x', 'rot_y', 'rot_z', 'sex']].to_numpy(np.float64), g['csv.values'], rtol=1e-10, atol=1e-12)
    # what the data loader and the model then read from these two files
    assert len(utils.get_xu_ranges([csv, csv])) == 6
