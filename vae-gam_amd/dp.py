"""Data parallelism over the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for the tests).

The reference has no distributed code.  The VAE-GAM step shards by minibatch, with
three batch-coupled pieces that keep global-batch semantics exact:
  * batch-norm statistics: per-layer all-reduce of the [sum, sumsq, count] (fwd) and
    [sum dy, sum dy*xhat] (bwd) triples -- a few hundred bytes each;
  * gains: covariates are all-gathered (B x C floats), every rank evaluates the full-batch GP /
    Cholesky with the same seeded noise and keeps its slice (the algebra is tiny and replicating
    it costs no bandwidth);
  * gradients: ONE all-reduce of the flat fp32 gradient buffer (+ one of the fp64 epsilon map),
    6.5 MB total, instead of 97 per-tensor calls.
"""
import os

import torch
import torch.distributed as dist


class DataParallelContext:
    def __init__(self, rank, world_size, device, group=None, noise_seed=1234):
        self.rank, self.world_size, self.device, self.group = rank, world_size, device, group
        self.bn_sync = _BnSync(self)
        self.noise_seed = noise_seed
        self._gens = {}
        # gloo cannot reduce device tensors in every build: stage through the host then (tests on one GPU)
        self._host_staging = dist.get_backend(group) == 'gloo' and device.type == 'cuda'
        t = torch.zeros(1, device=device)
        self.allreduce_sum_(t)                               # creates the communicator outside any graph capture

    def noise_generator(self, device):
        key = str(device)
        if key not in self._gens:
            g = torch.Generator(device=device)
            g.manual_seed(self.noise_seed)
            self._gens[key] = g
        return self._gens[key]

    @classmethod
    def from_env(cls, backend=None):
        rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])
        local = int(os.environ.get('LOCAL_RANK', '0'))
        cuda = torch.cuda.is_available()
        if backend is None:
            backend = os.environ.get('VG_DP_BACKEND') or ('nccl' if cuda else 'gloo')     # nccl = RCCL over xGMI
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if not dist.is_initialized():
            if cuda:
                local = local % max(torch.cuda.device_count(), 1)
                torch.cuda.set_device(local)
                if backend == 'nccl':
                    dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device('cuda', local))
                else:
                    dist.init_process_group(backend, rank=rank, world_size=world)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        return cls(rank, world, torch.device('cuda', local % max(torch.cuda.device_count(), 1)) if cuda else torch.device('cpu'))

    # ---- collectives
    def allreduce_sum_(self, t):
        if self._host_staging and t.is_cuda:
            h = t.cpu(); dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group); t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def allreduce_grads(self, flat_grads):
        """Sum the flat gradient buffers over ranks (loss terms are pre-scaled so that the SUM is the
        global-batch gradient): one collective per dtype buffer (fp32 parameters, fp64 epsilon map)."""
        for g in flat_grads:
            self.allreduce_sum_(g)

    def all_gather_rows(self, t):
        """(b, ...) per rank -> (world*b, ...) in rank order."""
        t = t.contiguous()
        if self._host_staging and t.is_cuda:
            parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(self.world_size)]
            dist.all_gather(parts, t.cpu(), group=self.group)
            return torch.cat(parts, 0).to(t.device)
        parts = [torch.empty_like(t) for _ in range(self.world_size)]
        dist.all_gather(parts, t, group=self.group)
        return torch.cat(parts, 0)

    def barrier(self):
        dist.barrier(group=self.group)

    def max_scalar(self, v):
        t = torch.tensor([float(v)], dtype=torch.float64, device='cpu' if self._host_staging else self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def sum_scalar_tensor(self, t):
        return self.allreduce_sum_(t.clone())

    def shard_loaders(self, loaders, global_batch, seed):
        return loaders            # file-backed loaders: every rank reads its own per-rank batch (see CLI)

    def shutdown(self):
        if dist.is_initialized():
            dist.destroy_process_group()


class _BnSync:
    """Callable handed to ops.bn_stats / bn_backward_: all-reduce of the raw double sums."""

    def __init__(self, ctx):
        self.ctx = ctx

    @property
    def world_size(self):
        return self.ctx.world_size

    def __call__(self, sums):
        return self.ctx.allreduce_sum_(sums)
