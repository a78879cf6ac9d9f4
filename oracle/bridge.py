"""TEST INFRASTRUCTURE: glue between the product model (vae_gam_amd.VAE) and the CPU oracle
(vaegam_oracle.py) -- copies a model's parameters into the oracle's name space so both run from
identical weights, and compares outputs.  Imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import vaegam_oracle as O  # noqa: E402

_KEYMAP = {'sa': 'sa', 'logstd': 'logstd', 'qu_m': 'qu_m', 'qu_S': 'qu_S', 'logkvar': 'logkvar', 'log_ls': 'log_ls', 'xu': 'xu'}


def oracle_config(model, glm_cdist=False):
    return O.OracleConfig(num_covariates=model.num_covariates, num_latents=model.num_latents,
                          num_inducing_pts=model.inducing_pts, gp_kl_scale=float(model.gp_kl_scale),
                          glm_reg_scale=float(model.glm_reg_scale), neural_covariates=bool(model.neural_covariates),
                          img=tuple(model.img_shape), nf=model.nf, lr=model.lr, glm_cdist=glm_cdist,
                          gp_jitter=float(getattr(model, 'gp_jitter', 0.0)))


def params_from_model(model):
    """Oracle parameter dict (CPU clones) from a product model."""
    p = {'epsilon': model.epsilon.detach().cpu().clone()}
    for cov, d in model.gp_params.items():
        for k, v in d.items():
            p['gp.%s.%s' % (cov, _KEYMAP[k])] = v.detach().cpu().clone()
    for name, layer in model._get_layers().items():
        p[name + '.weight'] = layer.weight.detach().cpu().clone()
        p[name + '.bias'] = layer.bias.detach().cpu().clone()
    return p


def model_param_by_oracle_name(model):
    out = {'epsilon': model.epsilon}
    for cov, d in model.gp_params.items():
        for k, v in d.items():
            if isinstance(v, torch.nn.Parameter):
                out['gp.%s.%s' % (cov, _KEYMAP[k])] = v
    for name, layer in model._get_layers().items():
        out[name + '.weight'] = layer.weight
        out[name + '.bias'] = layer.bias
    return out


def noise_to(noise, device):
    return {k: v.to(device) for k, v in noise.items()}


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
