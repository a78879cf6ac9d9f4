"""Fused Adam over flat parameter / gradient buffers.

`torch.optim.Adam(self.parameters(), lr)` in the reference (vae_reg_GP.py:179,429) walks 97
tensors; here all fp32 parameters live in ONE contiguous HBM buffer (and the fp64 epsilon map in
a second one), their `.grad`s are views of a matching flat gradient buffer, and the update is one
kernel launch per buffer (`vg_adam_step`).  The flat gradient buffer is also what the data-parallel
path all-reduces (one RCCL call instead of 97).  `state_dict()` / `load_state_dict()` speak
torch.optim.Adam's format so checkpoints stay interchangeable with the reference
(vae_reg_GP.py:457,480).
"""
import math
from typing import Dict, List, Sequence, Tuple

import torch

from . import ops


class FusedAdam:
    def __init__(self, named_params: Sequence[Tuple[str, torch.nn.Parameter]], lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 contiguous_groups: Sequence[Sequence[str]] = ()):
        """contiguous_groups: lists of parameter names laid out back to back (no padding between them) so that the model can
        view them as ONE stacked weight (the three encoder heads run as one GEMM / one batched GEMM).  The order inside
        the flat buffer is private: state_dict() indexes by position in `named_params`."""
        self.names = [n for n, _ in named_params]
        self.params: List[torch.nn.Parameter] = [p for _, p in named_params]
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.step_count = 0
        self.used = set(range(len(self.params)))          # indices that take part in training (get a state entry)
        self.groups: Dict[torch.dtype, dict] = {}
        device = self.params[0].device
        self.device = device
        for dtype in (torch.float32, torch.float64):
            idx = [i for i, p in enumerate(self.params) if p.dtype == dtype]
            if not idx:
                continue
            by_name = {self.names[i]: i for i in idx}
            packed, spans = [], {}
            for grp in contiguous_groups:
                if all(n in by_name for n in grp):
                    packed.append([by_name[n] for n in grp])
            in_pack = {i for g_ in packed for i in g_}
            order = [i for g_ in packed for i in g_] + [i for i in idx if i not in in_pack]
            last_of_group = {g_[-1] for g_ in packed}
            idx = order
            sizes = [self.params[i].numel() for i in idx]
            offs = [0]
            for i, s in zip(idx, sizes):
                nxt = offs[-1] + s
                if i not in in_pack or i in last_of_group:
                    nxt = ((nxt + 3) // 4) * 4                       # keep every free-standing view 16-byte aligned
                offs.append(nxt)
            total = offs[-1]
            for g_ in packed:
                k0 = idx.index(g_[0])
                spans[tuple(self.names[i] for i in g_)] = (offs[k0], sum(self.params[i].numel() for i in g_))
            self._spans = getattr(self, '_spans', {})
            self._spans.update({k: (dtype,) + v for k, v in spans.items()})
            flat_p = torch.zeros(total, dtype=dtype, device=device)
            flat_g = torch.zeros(total, dtype=dtype, device=device)
            for k, i in enumerate(idx):
                p = self.params[i]
                view = flat_p[offs[k]:offs[k] + sizes[k]].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = flat_g[offs[k]:offs[k] + sizes[k]].view(p.shape)
            self.groups[dtype] = dict(idx=idx, offs=offs, sizes=sizes, p=flat_p, g=flat_g,
                                      m=torch.zeros_like(flat_p), v=torch.zeros_like(flat_p))
        # [lr/(1-b1^t), sqrt(1-b2^t), t] ON THE DEVICE, advanced by a one-thread kernel inside the step (vg_adam_advance): a
        # host-staged buffer rewritten per step could be overwritten for step t+1 while step t's launches (or hipGraph
        # replays) that read it are still queued.  `step_count` is the host's mirror of t (checkpoints, state_dict).
        self._scalars = torch.zeros(3, dtype=torch.float64, device=device)

    def group_views(self, names):
        """(parameters, gradients) of a contiguous group as flat 1-D views of the flat buffers, or None."""
        sp = getattr(self, '_spans', {}).get(tuple(names))
        if sp is None:
            return None
        dtype, off, n = sp
        gr = self.groups[dtype]
        return gr['p'][off:off + n], gr['g'][off:off + n]

    # ---- torch.optim-like surface
    @property
    def param_groups(self):
        return [{'lr': self.lr, 'betas': self.betas, 'eps': self.eps, 'weight_decay': 0, 'amsgrad': False,
                 'params': self.params}]

    def flat_grads(self):
        return [g['g'] for g in self.groups.values()]

    def zero_grad(self, set_to_none: bool = False):
        for g in self.groups.values():
            g['g'].zero_()
        self._rebind_grads()

    def _rebind_grads(self):
        # autograd replaces .grad only if it was None; make sure the flat views are what it accumulates into
        for gr in self.groups.values():
            for k, i in enumerate(gr['idx']):
                p = self.params[i]
                if p.grad is None or p.grad.data_ptr() != gr['g'].data_ptr() + gr['offs'][k] * gr['g'].element_size():
                    p.grad = gr['g'][gr['offs'][k]:gr['offs'][k] + gr['sizes'][k]].view(p.shape)

    def advance(self):
        """t += 1 and the bias-correction scalars, on the device, as a launch of its own (captured with the step)."""
        ops.adam_advance_(self._scalars, self.lr, self.betas[0], self.betas[1])
        self.step_count += 1

    def set_step_count(self, t):
        """Resume / restore: put the device-side count at t (outside any capture)."""
        self.step_count = int(t)
        self._scalars.copy_(torch.tensor([0.0, 0.0, float(t)], dtype=torch.float64))

    def apply_update(self):
        for g in self.groups.values():
            ops.adam_step_(g['p'], g['g'], g['m'], g['v'], self.betas[0], self.betas[1], self.eps, self._scalars)

    def step(self):
        self.advance()
        self.apply_update()

    # ---- checkpoint format of torch.optim.Adam
    def _slot(self, i):
        p = self.params[i]
        gr = self.groups[p.dtype]
        k = gr['idx'].index(i)
        return gr, gr['offs'][k], gr['sizes'][k]

    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i in sorted(self.used):
                gr, off, n = self._slot(i)
                shape = self.params[i].shape
                state[i] = {'step': torch.tensor(float(self.step_count)),
                            'exp_avg': gr['m'][off:off + n].view(shape).clone(),
                            'exp_avg_sq': gr['v'][off:off + n].view(shape).clone()}
        group = {'lr': self.lr, 'betas': self.betas, 'eps': self.eps, 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'decoupled_weight_decay': False, 'params': list(range(len(self.params)))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        group = sd['param_groups'][0]
        if len(group['params']) != len(self.params):
            raise ValueError('optimizer state has %d parameters, model has %d' % (len(group['params']), len(self.params)))
        self.lr = float(group['lr']); self.betas = tuple(float(b) for b in group['betas']); self.eps = float(group['eps'])
        steps = set()
        for g in self.groups.values():
            g['m'].zero_(); g['v'].zero_()
        for pos, pid in enumerate(group['params']):
            st = sd['state'].get(pid)
            if st is None:
                continue
            gr, off, n = self._slot(pos)
            gr['m'][off:off + n].copy_(st['exp_avg'].reshape(-1).to(gr['m'].dtype))
            gr['v'][off:off + n].copy_(st['exp_avg_sq'].reshape(-1).to(gr['v'].dtype))
            steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError('per-parameter Adam step counts differ (%s): not representable by the fused optimiser' % sorted(steps))
        self.set_step_count(steps.pop() if steps else 0)
