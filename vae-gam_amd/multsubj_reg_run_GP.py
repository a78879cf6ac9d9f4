"""Command-line wrapper to train the VAE-GAM on MI355X (same flags as the reference's
multsubj_reg_run_GP.py:19-56; run as `python -m vae_gam_amd.multsubj_reg_run_GP ...`).

The post-hoc latent-UMAP / GP-plot / NIfTI reconstruction calls of the reference wrapper
(multsubj_reg_run_GP.py:83-92) are outside the hot path; `--recons_only` therefore only loads the
checkpoint and reports the test loss.  Multi-GPU: launch with torch.distributed.run, one process
per GPU; the wrapper picks up RANK / LOCAL_RANK / WORLD_SIZE and shards each minibatch.
"""
import argparse
import os
import time

import torch

from . import DataClass_GP as data
from . import vae_reg_GP as vae_reg
from .utils import str2bool


def build_parser():
    parser = argparse.ArgumentParser(description='user args for vae_gam model')
    parser.add_argument('--train_csv', type=str, metavar='N', default='', help='Full path to csv file with train dset.')
    parser.add_argument('--test_csv', type=str, metavar='N', default='', help='Full path to csv file with test dset.')
    parser.add_argument('--save_dir', type=str, metavar='N', default='', help='Dir where model params are saved to.')
    parser.add_argument('--batch-size', type=int, default=32, metavar='N', help='Input batch size for training (default: 32)')
    parser.add_argument('--epochs', type=int, default=300, metavar='N', help='Number of epochs to train (default: 300)')
    parser.add_argument('--seed', type=int, default=1, metavar='S', help='Random seed (default: 1)')
    parser.add_argument('--save_freq', type=int, default=100, metavar='N', help='How many epochs to wait before saving training status.')
    parser.add_argument('--test_freq', type=int, default=200, metavar='N', help='How many epochs to wait before testing.')
    parser.add_argument('--split', type=int, metavar='N', default=98, help='# of volumes per subject (latent plot colouring; kept for compatibility).')
    parser.add_argument('--glm_reg_scale', type=float, metavar='N', default=1.0, help='Scaling factor for GLM map regularization term (default: 1)')
    parser.add_argument('--glm_maps', type=str, metavar='N', default='', help='Path to csv file containing matrix with approximate GLM maps.')
    parser.add_argument('--num_inducing_pts', type=int, metavar='N', default=6, help='Number of inducing points for each regressor 1D GP.')
    parser.add_argument('--gp_kl_scale', type=float, metavar='N', default=10.0, help='Scaling factor for the gain KL terms.')
    parser.add_argument('--from_ckpt', type=str2bool, nargs='?', const=True, default=False, help='Start from a saved model state.')
    parser.add_argument('--ckpt_path', type=str, metavar='N', default='', help='Path to ckpt with saved model state to be loaded.')
    parser.add_argument('--recons_only', type=str2bool, nargs='?', const=True, default=False, help='Skip training.')
    parser.add_argument('--neural_covariates', type=str2bool, nargs='?', const=True, default=True,
                        help='Covariate set includes neural/biological effects to be convolved with the HRF.')
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    torch.manual_seed(args.seed)
    if args.save_dir == '':
        args.save_dir = os.getcwd()
    if not os.path.exists(args.save_dir):
        os.makedirs(args.save_dir, exist_ok=True)
    main_start = time.time()
    dp = None
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        from . import dp as dpmod
        dp = dpmod.DataParallelContext.from_env()
    per_rank = args.batch_size if dp is None else args.batch_size // dp.world_size
    loaders_dict = data.setup_data_loaders(batch_size=per_rank, train_csv=args.train_csv, test_csv=args.test_csv)
    if dp is not None:
        loaders_dict = dp.shard_loaders(loaders_dict, args.batch_size, args.seed)
    model = vae_reg.VAE(num_inducing_pts=args.num_inducing_pts, gp_kl_scale=args.gp_kl_scale,
                        glm_reg_scale=args.glm_reg_scale, glm_maps=args.glm_maps, save_dir=args.save_dir,
                        csv_files=[args.train_csv, args.test_csv], neural_covariates=args.neural_covariates,
                        data_parallel=dp)
    if args.from_ckpt:
        assert os.path.exists(args.ckpt_path), 'Oops, looks like ckpt file given does NOT exist!'
        print('=' * 40)
        print('Loading model state from: {}'.format(args.ckpt_path))
        model.load_state(filename=args.ckpt_path)
    if not args.recons_only:
        model.train_loop(loaders_dict, epochs=args.epochs, test_freq=args.test_freq, save_freq=args.save_freq,
                         save_dir=args.save_dir)
    else:
        assert args.from_ckpt, 'To choose recons_only option, --from_ckpt needs to be TRUE.'
        model.test_epoch(loaders_dict['test'])
    print('Total model runtime (seconds): {}'.format(time.time() - main_start))
    return model


if __name__ == "__main__":
    main()
