"""The product model (vae_gam_amd.VAE: autograd nodes, batched decoder, fused GAM/ELBO, fused Adam) driven through
the host build of the HIP kernels on a 21x21x21 toy geometry, against the CPU oracle.  No GPU needed; the
-m gpu tests repeat this at 41x49x35 on the real library."""
import numpy as np
import pytest
import torch

import bridge
import toy_case as T
import vaegam_oracle as O
from vae_gam_amd import _lib


@pytest.fixture(scope='module', autouse=True)
def emu():
    prev = _lib._LIB
    T.load_emu_library()
    yield
    _lib._LIB = prev


def test_train_step_matches_oracle_toy_geometry():
    B, C = 4, 3
    x, cov, xu, glm = T.make_inputs(B, C, seed=5)
    model = T.make_model(C, xu, glm)
    cfg = bridge.oracle_config(model)
    params = bridge.params_from_model(model)
    noise = O.draw_noise(B, cfg, torch.Generator().manual_seed(9))
    opt = O.AdamState(lr=cfg.lr)
    out, grads = O.train_step(params, opt, cfg, x, cov, torch.from_numpy(glm), noise)
    ids = torch.zeros(B, dtype=torch.int64)
    loss = model.train_step(ids, cov, x, noise=noise)
    np.testing.assert_allclose(loss.numpy(), out['loss'].detach().numpy(), rtol=1e-4)
    byname = bridge.model_param_by_oracle_name(model)
    for k, g in grads.items():
        if g is None or k.endswith(('.logkvar', '.log_ls')):
            continue
        a = byname[k].grad.double().flatten().numpy(); b = g.double().flatten().numpy()
        nb = np.sqrt((b * b).sum())
        assert np.sqrt(((a - b) ** 2).sum()) <= 5e-3 * nb + 1e-6, (k, np.sqrt(((a - b) ** 2).sum()), nb)
    for k, p in byname.items():                    # parameters after the fused Adam step
        if k.endswith(('.logkvar', '.log_ls')) or grads.get(k) is None:
            continue
        # the first Adam step moves every entry by ~lr*sign(g): compare where the sign of g is not rounding noise
        gref = grads[k].double().flatten().numpy()
        sel = np.abs(gref) > 1e-3 * max(np.abs(gref).max(), 1e-30)
        np.testing.assert_allclose(p.detach().double().flatten().numpy()[sel], params[k].double().flatten().numpy()[sel],
                                   atol=3e-5, err_msg=k)


def test_return_latent_rec_keys_and_shapes():
    B, C = 2, 3
    x, cov, xu, glm = T.make_inputs(B, C, seed=6)
    model = T.make_model(C, xu, glm)
    with torch.no_grad():
        loss, z, imgs = model.forward(torch.zeros(B, dtype=torch.int64), cov, x, 'test', return_latent_rec=True, train_mode=False)
    assert loss.shape == (1,) and z.shape == (B, 32)
    assert list(imgs.keys()) == ['base', 'task', 'x_mot', 'y_mot', 'z_mot', 'pitch_mot', 'roll_mot', 'yaw_mot', 'sex', 'full_rec']
    for k in ('base', 'task', 'x_mot', 'y_mot', 'full_rec'):
        assert imgs[k].shape == (B, 21 * 21 * 21)
    for k in ('z_mot', 'sex'):
        assert imgs[k] == {}
    np.testing.assert_allclose(imgs['full_rec'], imgs['base'] + imgs['task'] + imgs['x_mot'] + imgs['y_mot'], atol=1e-5)
