"""TEST INFRASTRUCTURE (bench.py's cpu_baseline leg only): the host-side work the reference's `VAE.forward`
performs on EVERY training call besides the arithmetic -- emulated so that the CPU baseline can also be quoted
"as shipped" (SURVEY 8d, H7).  Nothing under vae-gam_amd/ imports this.

What the reference does per forward with `train_mode=True` (vae_reg_GP.py):
  * `.detach().cpu().numpy()` of the base map, every effect map and the full reconstruction (:331, :391, :392):
    (C+2) arrays of B x V floats (a device-to-host copy when the model sits on a GPU; a materialised host array here);
  * `utils.log_map` for slices 12, 15, 18 of the base map, the task map and the full reconstruction (:333-337, :381-386,
    :395-398 -> utils.py:373-389): 9 x B slices, each rotated by 90 degrees with scipy.ndimage.rotate and handed to
    SummaryWriter.add_image (which scales to uint8 and PNG-encodes it);
  * `utils.log_beta` per covariate (:370-372 -> utils.py:347-371): a pandas frame sorted by the covariate, one matplotlib
    figure (line + fill_between + legend) handed to SummaryWriter.add_figure (which renders it to an RGB buffer).
tensorboard is not installed in this image: add_image / add_figure are replaced by what they cost on the host -- the uint8
scaling + zlib deflate of a PNG encode, and an Agg canvas draw + RGB buffer read-back + deflate.
"""
import time
import zlib

import numpy as np


def _add_image(slc):
    a = np.asarray(slc, dtype=np.float32)
    u8 = (np.clip(a, 0.0, 1.0) * 255.0).astype(np.uint8)          # tensorboard's make_np / image scaling
    return len(zlib.compress(u8.tobytes(), 6))                     # PNG = filtered rows + deflate


def _log_map(m, img_shape, sl, B):
    from scipy import ndimage
    m = m.reshape((B,) + tuple(img_shape))
    n = 0
    for i in range(B):
        n += _add_image(ndimage.rotate(m[i, sl, :, :], 90))        # utils.py:386-389
    return n


def _log_beta(xq, beta_mean, beta_cov, name):
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    import pandas as pd
    two_sigma = 2 * np.sqrt(np.abs(np.diag(beta_cov)))
    df = pd.DataFrame.from_dict({'xq': xq, 'mean': beta_mean, 'two_sig': two_sigma}).sort_values(by=['xq'])
    fig = plt.figure()
    plt.plot(df['xq'], df['mean'], c='darkblue', alpha=0.5, label='Beta posterior mean')
    plt.fill_between(df['xq'], df['mean'] - df['two_sig'], df['mean'] + df['two_sig'], color='lightblue', alpha=0.3, label='2 sigma')
    plt.legend(loc='best'); plt.title('Beta_%s' % name); plt.xlabel('Covariate'); plt.ylabel('Beta Ouput')
    fig.canvas.draw()                                              # add_figure -> figure_to_image
    buf = np.asarray(fig.canvas.buffer_rgba())
    n = len(zlib.compress(buf[..., :3].tobytes(), 6))
    plt.close(fig)
    return n


def time_forward_logging(out, covariates, cfg):
    """Seconds of host work one training forward of the reference adds on top of the arithmetic, measured on the oracle's own
    outputs `out` (vaegam_oracle.forward) for covariates (B, C)."""
    B = covariates.shape[0]
    names = [c.name for c in cfg.schema]
    t0 = time.perf_counter()
    maps = out['maps']
    host = {k: np.array(v.detach().cpu().numpy(), copy=True) for k, v in maps.items()}       # the (C+2) B x V host arrays
    for key in ('base', names[0], 'full_rec'):
        for sl in (12, 15, 18):
            _log_map(host[key], cfg.img, sl, B)
    cov = covariates.detach().cpu().numpy()
    for i, n in enumerate(names):
        _log_beta(cov[:, i], out['beta_mean'][n].detach().cpu().numpy(), out['beta_cov'][n].detach().cpu().numpy(), n)
    return time.perf_counter() - t0
