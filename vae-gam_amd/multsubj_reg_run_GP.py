"""Command-line wrapper to train the VAE-GAM on MI355X (same flags as the reference's
multsubj_reg_run_GP.py:19-56; run as `python -m vae_gam_amd.multsubj_reg_run_GP ...`).

After training -- or straight from a checkpoint with `--recons_only` -- the wrapper runs the reference's export pipeline
(multsubj_reg_run_GP.py:79-92): `model.plot_GPs` (per-covariate GP posterior CSVs), `recon.mk_single_volumes` (one NIfTI
per volume and map) and `recon.mk_avg_maps(mk_motion_maps=True)` (subject and grand averages), all on the training set as
the reference does.  `project_latent` (a UMAP scatter plot of the latent means; umap-learn / plotting) is not part of the
hot path and is skipped with a printed note.

Multi-GPU: launch with torch.distributed.run, one process per GPU; the wrapper picks up RANK / LOCAL_RANK / WORLD_SIZE,
`--batch-size` stays the GLOBAL minibatch and every rank draws its own slice of it (dp.ShardedBatchSampler); checkpoints
and the export pipeline run on rank 0 only.
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
    parser.add_argument('--gp_jitter', type=float, metavar='N', default=0.0,
                        help='(extension) Ku + jitter*I in the GP posteriors; 0 = the reference\'s plain inverse. Needed for dense inducing grids '
                             '(e.g. --num_inducing_pts 64: 1e-4), where Ku is singular in any precision.')
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
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 or os.environ.get('VG_DP_FORCE') == '1':
        from . import dp as dpmod
        dp = dpmod.DataParallelContext.from_env()
    rank = 0 if dp is None else dp.rank
    if dp is not None and args.batch_size % dp.world_size:
        raise SystemExit('--batch-size %d (the GLOBAL minibatch) must be a multiple of the %d ranks' % (args.batch_size, dp.world_size))
    # the reference's loaders (whole data set, GLOBAL minibatch, multsubj_reg_run_GP.py:69): what the export pipeline iterates -- gains
    # (joint draw, HRF along the batch) and the decoder's batch statistics depend on the batch composition, so the exported maps must
    # see the batches a single process would; under data parallelism the train / test loops get re-built loaders that hand each rank
    # its slice of every global minibatch (they only take the data sets from these)
    # (single process on a GPU: minibatches are staged in pinned memory and copied one batch ahead on a side stream)
    full_loaders = data.setup_data_loaders(batch_size=args.batch_size, train_csv=args.train_csv, test_csv=args.test_csv,
                                           prefetch_device='cuda' if (dp is None and torch.cuda.is_available()) else None)
    loaders_dict = full_loaders if dp is None else dp.shard_loaders(full_loaders, args.batch_size, args.seed)
    model = vae_reg.VAE(num_inducing_pts=args.num_inducing_pts, gp_kl_scale=args.gp_kl_scale,
                        glm_reg_scale=args.glm_reg_scale, glm_maps=args.glm_maps, save_dir=args.save_dir,
                        csv_files=[args.train_csv, args.test_csv], neural_covariates=args.neural_covariates,
                        data_parallel=dp, dp_gain=os.environ.get('VG_DP_GAIN', 'global'), gp_jitter=args.gp_jitter)
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
    export_outputs(model, full_loaders, args, dp)
    if rank == 0:
        print('Total model runtime (seconds): {}'.format(time.time() - main_start))
    return model


def export_outputs(model, loaders_dict, args, dp=None):
    """The post-training block of the reference wrapper (multsubj_reg_run_GP.py:83-86 and, for --recons_only, :89-92), on
    the training set.  Rank 0 only under data parallelism: the replicas are identical and the export runs single-process.  The
    process group is torn down FIRST (after one barrier): the other ranks return instead of sitting in a collective for as long as
    the export takes (file output of every volume x map: minutes on a real data set, past the RCCL watchdog's patience)."""
    from . import build_model_recons as recon
    if dp is not None:
        dp.barrier()
        dp.shutdown()
        if dp.rank != 0:
            return
    saved_dp, model.dp = model.dp, None
    try:
        print('project_latent (UMAP plot of the latent space, vae_reg_GP.py:542-583) is not part of this build: skipped.')
        model.plot_GPs(csv_file=args.train_csv, save_dir=args.save_dir)
        recon.mk_single_volumes(loaders_dict['UnShuffled_train'], model, args.train_csv, args.save_dir)
        recon.mk_avg_maps(args.train_csv, model, args.save_dir, mk_motion_maps=True)
    finally:
        model.dp = saved_dp


if __name__ == "__main__":
    main()
