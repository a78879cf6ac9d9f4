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
        """The DataLoaders of DataClass_GP.setup_data_loaders re-built so that every rank draws ITS slice of every global
        minibatch: same data sets, `ShardedBatchSampler` instead of the per-process shuffling (which, seeded identically on
        every rank, would hand all ranks the same samples)."""
        from torch.utils.data import DataLoader
        out = {}
        for name, ld in loaders.items():
            sampler = ShardedBatchSampler(len(ld.dataset), global_batch, self.rank, self.world_size,
                                          shuffle=(name == 'Shuffled_train'), seed=seed)
            out[name] = DataLoader(ld.dataset, batch_sampler=sampler, num_workers=0, collate_fn=ld.collate_fn)
        return out

    def shutdown(self):
        if dist.is_initialized():
            dist.destroy_process_group()


class ShardedBatchSampler:
    """batch_sampler for torch's DataLoader under data parallelism (the reference has one process, DataClass_GP.py:73-89).

    Every rank builds the SAME order of the data set for epoch e -- a permutation seeded with (seed, e) when shuffling, the
    identity otherwise --, cuts it into global minibatches of `global_batch` indices and keeps the contiguous slice
    [rank*b, (rank+1)*b) of each, b = global_batch / world: the ranks' slices are disjoint and their union, in rank order,
    is the minibatch a single process would have drawn (the layout DeviceResidentData and forward_core's all-gather assume).
    The last, short minibatch (the reference keeps it, drop_last=False) is cut to a multiple of the world size so that all
    ranks take the same number of steps with equal-sized batches -- the collectives inside a step would otherwise deadlock;
    at most world-1 samples per epoch are left out.  The epoch advances by itself on every __iter__ (one pass = one epoch on
    every rank); set_epoch() pins it."""

    def __init__(self, n, global_batch, rank, world, shuffle, seed=0):
        if global_batch % world:
            raise ValueError('global batch %d is not a multiple of the world size %d' % (global_batch, world))
        self.n, self.global_batch, self.rank, self.world, self.shuffle, self.seed = int(n), int(global_batch), rank, world, shuffle, int(seed)
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _batches(self, epoch):
        if self.shuffle:
            g = torch.Generator(); g.manual_seed(self.seed * 1000003 + epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        for s in range(0, self.n, self.global_batch):
            glob = order[s:s + self.global_batch]
            b = len(glob) // self.world
            if b == 0:
                return
            yield glob[self.rank * b:(self.rank + 1) * b]

    def __iter__(self):
        e = self.epoch
        self.epoch += 1
        return self._batches(e)

    def __len__(self):
        full, rest = divmod(self.n, self.global_batch)
        return full + (1 if rest // self.world > 0 else 0)


class _BnSync:
    """Callable handed to ops.bn_stats / bn_backward_: all-reduce of the raw double sums."""

    def __init__(self, ctx):
        self.ctx = ctx

    @property
    def world_size(self):
        return self.ctx.world_size

    def __call__(self, sums):
        return self.ctx.allreduce_sum_(sums)
