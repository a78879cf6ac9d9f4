"""Tensor-level operators over the HIP kernels (C ABI: include/vaegam.h).

Every function takes torch tensors that already live where the library can reach them
(HBM for libvaegam_hip.so), passes raw pointers + the current HIP stream, and allocates
outputs/workspace through torch's caching allocator (so the whole train step can be captured
into a hipGraph).  `BnConvAct` and `GamElbo` are the autograd nodes the model is built from;
their backward passes are explicit kernel sequences, not autograd traces.

Storage convention: a layer stores its PRE-activation output; the consumer applies the ReLU
and the batch-norm affine while loading (`relu_in`, `gamma`/`beta`).  This removes every
stand-alone ReLU / batch-norm-apply pass of the reference graph (vae_reg_GP.py:238-264).
"""
import ctypes
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ConvDesc, WgradDesc

BN_EPS = 1e-5            # nn.BatchNorm3d default (vae_reg_GP.py:194-196)

# Optional per-launch timing with HIP events recorded on the stream the kernels are launched on
# (bench.py's roofline leg).  PROFILE maps "<entry point>:<layer>" -> list of (start, end) events.
PROFILE = None
_LABEL = ['']


class label:
    """with ops.label('convt5'): ... tags the launches inside (profiling only)."""
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.prev = _LABEL[0]; _LABEL[0] = self.name

    def __exit__(self, *a):
        _LABEL[0] = self.prev


def _call(t, fn, *args):
    """Launch one C-ABI entry point on t's current stream (optionally bracketed by HIP events)."""
    lib = _lib.get_lib()
    if not t.is_cuda and not lib.host_pointers_ok:
        raise RuntimeError('refusing to launch %s on a non-GPU tensor: the HIP kernels need device pointers (no CPU path)' % fn)
    if PROFILE is not None and t.is_cuda:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.call(fn, *args, _stream(t))
        e1.record()
        PROFILE.setdefault('%s:%s' % (fn, _LABEL[0]), []).append((e0, e1))
    else:
        lib.call(fn, *args, _stream(t))


# Parameter-gradient kernels (weight gradient, bias sum) of a layer depend on dy but nothing downstream depends
# on them until the optimiser: they run on a second HIP stream beside the data-gradient chain (which is what the
# next layer's backward waits for).  Round 1 measured a LOSS (7.75 vs 7.29 ms/step at batch 32: the persistent
# weight-gradient blocks of that time held the LDS of every CU, so the chains did not overlap and the extra joins only
# added boundaries).  With round 2's kernels (2-3 weight-gradient blocks per CU at 56 KB, matrix-core data gradients)
# the two chains do overlap: 8.10 -> 7.85 ms/step at batch 64 / 8 covariates, 3.20 -> 3.03 at batch 32 / 3.  On.
import os as _os
SIDE_STREAM = bool(int(_os.environ.get('VG_SIDE_STREAM', '1')))
FUSE_LAST_BN_BWD = bool(int(_os.environ.get('VG_FUSE_LAST_BN_BWD', '1')))     # bnt5 backward + convt5 data gradient in two fused passes
WGRAD_BN_SUMS = bool(int(_os.environ.get('VG_WGRAD_BN_SUMS', '1')))          # bnt5's backward reductions from convt5's grouped weight gradient (no reduce pass)
FC_SIDE_STREAM = bool(int(_os.environ.get('VG_FC_SIDE_STREAM', '0')))     # fully connected dW/db on the second stream: measured 4.41 vs 4.28 ms/step (worse), off
_SIDE = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


class on_side_stream:
    """with on_side_stream(ref, tensors...): launches inside go to the side stream, after everything already
    queued on the current stream; `tensors` are marked as used there (allocator safety)."""
    def __init__(self, ref, *tensors, enabled=None):
        self.active = bool((SIDE_STREAM if enabled is None else enabled) and ref.is_cuda)
        self.ref, self.tensors = ref, tensors

    def __enter__(self):
        if not self.active:
            return self
        self.side = _side_stream(self.ref.device)
        self.side.wait_stream(torch.cuda.current_stream(self.ref.device))
        _queue_join(self.ref.device)                     # the stream that ran backward() waits for the side work at its end
        for t in self.tensors:
            if t is not None:
                t.record_stream(self.side)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self.active:
            self.ctx.__exit__(*a)


_JOIN_QUEUED = set()


def _queue_join(device):
    key = (device.type, device.index)
    if key in _JOIN_QUEUED:
        return
    _JOIN_QUEUED.add(key)

    def _cb():
        _JOIN_QUEUED.discard(key)
        join_side_stream(device)
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_cb)     # runs when this backward pass has finished
    except RuntimeError:                                                  # not inside a backward pass
        _JOIN_QUEUED.discard(key)


def join_side_stream(device):
    """The current stream waits for the parameter-gradient kernels queued on the side stream."""
    key = (device.type, device.index)
    if key in _SIDE:
        torch.cuda.current_stream(device).wait_stream(_SIDE[key])


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t):
    if t.is_cuda:
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return None


def _chk(t, dtype=torch.float32):
    assert t.is_contiguous(), 'tensor must be contiguous'
    assert t.dtype == dtype, 'expected %s, got %s' % (dtype, t.dtype)
    return t


# --------------------------------------------------------------------------- layer geometry
@dataclass(frozen=True)
class ConvSpec:
    """One nn.Conv3d / nn.ConvTranspose3d of vae_reg_GP.py:189-215."""
    kind: str                      # 'conv' | 'convt'
    ci: int
    co: int
    k: Tuple[int, int, int]
    stride: int
    pad: Tuple[int, int, int] = (0, 0, 0)
    outpad: Tuple[int, int, int] = (0, 0, 0)
    name: str = ''

    def out_size(self, i):
        if self.kind == 'conv':
            return tuple((i[a] + 2 * self.pad[a] - self.k[a]) // self.stride + 1 for a in range(3))
        return tuple((i[a] - 1) * self.stride - 2 * self.pad[a] + self.k[a] + self.outpad[a] for a in range(3))


def pack_weight(w: torch.Tensor, spec: ConvSpec, direction: str) -> torch.Tensor:
    """Repack a layer weight into the [ci_launch][tap][co_launch] order the kernels read through
    the scalar path.  direction: 'fwd' (layer forward) or 'bwd' (data gradient)."""
    fwd = direction == 'fwd'
    if (spec.kind == 'conv') == fwd:
        # conv forward / convT data-gradient: strided correlation with the weight as stored
        # conv w[co][ci][k] -> [ci][k][co];   convT w[ci][co][k] (launch ci=co_l, co=ci_l) -> [co_l][k][ci_l]
        return w.permute(1, 2, 3, 4, 0).contiguous()
    # convT forward / conv data-gradient: transposed form.  stride 1 runs as a correlation with the
    # kernel flipped; stride 2 runs in gather form with the kernel as stored.
    if spec.stride == 1:
        w = w.flip(2, 3, 4)
    return w.permute(0, 2, 3, 4, 1).contiguous()


def pack_mode(spec: ConvSpec, direction: str) -> int:
    """vg_pack_weights mode producing the same image as pack_weight(w, spec, direction)."""
    if (spec.kind == 'conv') == (direction == 'fwd'):
        return 0
    return 2 if spec.stride == 1 else 1


class PackedWeights:
    """All packed conv-weight images of a model in one buffer, refreshed by ONE launch per step."""

    def __init__(self, layers, flat, offsets, sizes=None):
        """layers: [(name, spec, weight Parameter)], flat: the flat fp32 parameter buffer, offsets: {name: element offset},
        sizes: {name: (input spatial size, output spatial size)} -- with it the matrix-core plans (mm_plan) of every layer are
        built too and their A images are gathered from the flat buffer by one more launch per step."""
        self.mm = {}                                  # (name, direction) -> (MmPlan, A-image view)
        self._mm_idx = None
        if sizes is not None and USE_MM:
            import numpy as np
            idx_all, spans, pos = [], {}, 0
            for name, spec, w in layers:
                isz, osz = sizes[name]
                for direction in ('fwd', 'bwd'):
                    if direction == 'bwd' and name == 'conv1':
                        continue
                    plan = mm_plan(spec, 'fwd', isz) if direction == 'fwd' else mm_plan(spec, 'bwd', osz, isz)
                    if plan is None:
                        continue
                    gi = np.where(plan.aidx >= 0, plan.aidx + offsets[name], -1).astype(np.int32)
                    idx_all.append(gi); spans[(name, direction)] = (plan, pos, gi.size); pos += gi.size
            if idx_all:
                self._mm_idx = torch.from_numpy(np.concatenate(idx_all)).to(flat.device)
                self._mm_buf = torch.zeros(pos, dtype=torch.float32, device=flat.device)
                self.mm = {k: (plan, self._mm_buf[o:o + n]) for k, (plan, o, n) in spans.items()}
        segs, views, dst = [], {}, 0
        for name, spec, w in layers:
            for direction in ('fwd', 'bwd'):
                d0, d1 = w.shape[0], w.shape[1]
                kvol = w[0, 0].numel()
                n = w.numel()
                segs.append([offsets[name], dst, d0, d1, kvol, pack_mode(spec, direction), n, 0])
                views[(name, direction)] = (dst, n, pack_weight(w.detach(), spec, direction).shape)
                dst += ((n + 3) // 4) * 4
        self.total = dst
        self.flat = flat
        self.buf = torch.empty(dst, dtype=torch.float32, device=flat.device)
        self.segs = torch.tensor(segs, dtype=torch.int64).to(flat.device)
        self.nseg = len(segs)
        self._views = {k: self.buf[o:o + n].view(shape) for k, (o, n, shape) in views.items()}
        # the kernel walks every destination element, including the alignment gaps: give the last segment the tail
        self.elems = dst

    def refresh(self):
        _call(self.flat, 'vg_pack_weights', _p(self.flat), _p(self.buf), _p(self.segs), self.nseg, self.elems)
        if self._mm_idx is not None:
            _call(self.flat, 'vg_gather_f32', _p(self.flat), _p(self._mm_idx), _p(self._mm_buf), self._mm_idx.numel())

    def get_mm(self, name, direction):
        return self.mm.get((name, direction))

    def get(self, name, direction):
        return self._views[(name, direction)]


def _conv_desc(N, ci, co, isz, osz, k, stride, pad, relu_in, per_group):
    return ConvDesc(N, ci, co, isz[0], isz[1], isz[2], osz[0], osz[1], osz[2], k[0], k[1], k[2], stride,
                    pad[0], pad[1], pad[2], int(relu_in), int(per_group))


def conv_forward(x, wpk, bias, spec: ConvSpec, relu_in=False, scale=None, shift=None, per_group=1, next_bn=None):
    """Layer forward: y = conv/convT(P(x)) + bias, y is the pre-activation.
    next_bn = per_group of the BatchNorm3d that consumes relu(y): the stride-2 transposed-conv kernel then also accumulates
    the statistics partials for it (no separate pass over y) and the call returns (y, partials) -- the caller hands the
    partials to the consuming layer's bn_stats(pre=...) explicitly."""
    lib = _lib.get_lib()
    N = x.shape[0]
    isz = tuple(x.shape[2:]); osz = spec.out_size(isz)
    y = torch.empty((N, spec.co) + osz, dtype=torch.float32, device=x.device)
    _chk(x); _chk(wpk)
    if spec.kind == 'conv':
        assert spec.pad == (0, 0, 0)
        d = _conv_desc(N, spec.ci, spec.co, isz, osz, spec.k, spec.stride, (0, 0, 0), relu_in, per_group)
        _call(x, 'vg_corr3d', ctypes.byref(d), _p(x), _p(wpk), _p(bias), _p(scale), _p(shift), None, _p(y))
    elif spec.stride == 1:
        padc = tuple(spec.k[a] - 1 - spec.pad[a] for a in range(3))
        d = _conv_desc(N, spec.ci, spec.co, isz, osz, spec.k, 1, padc, relu_in, per_group)
        _call(x, 'vg_corr3d', ctypes.byref(d), _p(x), _p(wpk), _p(bias), _p(scale), _p(shift), None, _p(y))
    else:
        d = _conv_desc(N, spec.ci, spec.co, isz, osz, spec.k, 2, spec.pad, relu_in, per_group)
        if next_bn:
            chunks = lib.size('vg_tconv3d_s2_stats_chunks', ctypes.byref(d), int(next_bn))
            G = N // int(next_bn)
            part = torch.empty(G * spec.co * chunks * 2, dtype=torch.float64, device=x.device)
            _call(x, 'vg_tconv3d_s2_stats', ctypes.byref(d), _p(x), _p(wpk), _p(bias), _p(scale), _p(shift), _p(y), int(next_bn), 1,
                     _p(part))
            return y, part
        else:
            _call(x, 'vg_tconv3d_s2', ctypes.byref(d), _p(x), _p(wpk), _p(bias), _p(scale), _p(shift), None, _p(y))
    if next_bn:
        raise ValueError('next_bn: only the stride-2 transposed-conv kernel accumulates the next layer\'s statistics')
    return y


def conv_backward_data(dy, wpk_bwd, spec: ConvSpec, in_size, mask_src=None):
    """Gradient w.r.t. the layer's (post-prologue) input; `mask_src` fuses the producer's ReLU backward."""
    lib = _lib.get_lib()
    N = dy.shape[0]
    osz = tuple(dy.shape[2:]); isz = tuple(in_size)
    dx = torch.empty((N, spec.ci) + isz, dtype=torch.float32, device=dy.device)
    _chk(dy); _chk(wpk_bwd)
    if spec.kind == 'convt':
        # dx[i] = sum_k dy[i*s - pad + k] w[k]: strided correlation over dy
        d = _conv_desc(N, spec.co, spec.ci, osz, isz, spec.k, spec.stride, spec.pad, False, 1)
        _call(dy, 'vg_corr3d', ctypes.byref(d), _p(dy), _p(wpk_bwd), None, None, None, _p(mask_src), _p(dx))
    elif spec.stride == 1:
        padc = tuple(spec.k[a] - 1 for a in range(3))
        d = _conv_desc(N, spec.co, spec.ci, osz, isz, spec.k, 1, padc, False, 1)
        _call(dy, 'vg_corr3d', ctypes.byref(d), _p(dy), _p(wpk_bwd), None, None, None, _p(mask_src), _p(dx))
    else:
        d = _conv_desc(N, spec.co, spec.ci, osz, isz, spec.k, 2, (0, 0, 0), False, 1)
        _call(dy, 'vg_tconv3d_s2', ctypes.byref(d), _p(dy), _p(wpk_bwd), None, None, None, _p(mask_src), _p(dx))
    return dx


def conv_weight_grad(x, dy, spec: ConvSpec, relu_in=False, scale=None, shift=None, per_group=1, out=None):
    """dL/dW in the layer's own weight layout; the prologue of the forward is re-applied to x on load."""
    lib = _lib.get_lib()
    N = x.shape[0]
    isz = tuple(x.shape[2:]); osz = tuple(dy.shape[2:])
    _chk(x); _chk(dy)
    if spec.kind == 'conv':
        d = WgradDesc(N, spec.co, spec.ci, *osz, *isz, *spec.k, spec.stride, 0, 0, 0, 1, int(relu_in), int(per_group))
        a, b = x, dy
        shape = (spec.co, spec.ci) + tuple(spec.k)
    else:
        d = WgradDesc(N, spec.ci, spec.co, *isz, *osz, *spec.k, spec.stride, *spec.pad, 0, int(relu_in), int(per_group))
        a, b = dy, x
        shape = (spec.ci, spec.co) + tuple(spec.k)
    nbytes = lib.size('vg_wgrad3d_ws_bytes', ctypes.byref(d))
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x.device)
    if out is not None:          # accumulate straight into the parameter's .grad (a view of the flat gradient buffer)
        assert out.shape == shape and out.is_contiguous() and out.dtype == torch.float32
        _call(x, 'vg_wgrad3d', ctypes.byref(d), _p(a), _p(b), _p(scale), _p(shift), _p(ws), _p(out), 1)
        return None
    dw = torch.empty(shape, dtype=torch.float32, device=x.device)
    _call(x, 'vg_wgrad3d', ctypes.byref(d), _p(a), _p(b), _p(scale), _p(shift), _p(ws), _p(dw), 0)
    return dw


# --------------------------------------------------------------------------- batch norm pieces
def _bn_ws(x, N, C, P, per_group):
    nbytes = _lib.get_lib().size('vg_bn_ws_bytes', N, C, P, per_group)
    return torch.empty(nbytes // 8, dtype=torch.float64, device=x.device)


def bn_stats(x, gamma, beta, relu, per_group, sync=None, pre=None):
    """Batch statistics of relu?(x) per (group, channel) -> (scale, shift, mean, rstd), each [G*C].
    `sync(t)` (optional) all-reduces the raw [sum, sumsq, count] triples across data-parallel ranks.
    `pre`: the partials of relu(x) with THIS per_group that the kernel which wrote x accumulated
    (conv_forward(..., next_bn=per_group)); x is then not read at all."""
    lib = _lib.get_lib()
    N, C = x.shape[0], x.shape[1]
    P = x[0, 0].numel()
    G = N // per_group
    out = torch.empty((4, G * C), dtype=torch.float32, device=x.device)
    if pre is not None:
        assert relu, 'producer-side partials are statistics of relu(x)'
        part = pre
        assert part.dtype == torch.float64 and part.device == x.device and part.numel() % (G * C * 2) == 0
        chunks = part.numel() // (G * C * 2)
        sums = torch.empty((G * C, 3), dtype=torch.float64, device=x.device)
        count = float(per_group * P)
        if sync is None:
            _call(x, 'vg_bn_stats_from_parts', _p(part), G, C, chunks, count, _p(gamma), _p(beta), BN_EPS, None, _p(sums),
                     _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]))
        else:
            _call(x, 'vg_bn_stats_from_parts', _p(part), G, C, chunks, count, _p(gamma), _p(beta), BN_EPS, _p(sums), None,
                     None, None, None, None)
            sums = sync(sums)
            _call(x, 'vg_bn_finalize', _p(sums), G, C, _p(gamma), _p(beta), BN_EPS, _p(out[0]), _p(out[1]), _p(out[2]),
                     _p(out[3]))
        return out[0], out[1], out[2], out[3]
    ws = _bn_ws(x, N, C, P, per_group)
    if sync is None:
        _call(x, 'vg_bn_stats', _p(_chk(x)), N, C, P, per_group, int(relu), _p(gamma), _p(beta), BN_EPS, _p(ws), None,
                 _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]))
    else:
        sums = torch.empty((G * C, 3), dtype=torch.float64, device=x.device)
        _call(x, 'vg_bn_stats', _p(_chk(x)), N, C, P, per_group, int(relu), _p(gamma), _p(beta), BN_EPS, _p(ws), _p(sums),
                 None, None, None, None)
        sums = sync(sums)
        _call(x, 'vg_bn_finalize', _p(sums), G, C, _p(gamma), _p(beta), BN_EPS, _p(out[0]), _p(out[1]), _p(out[2]),
                 _p(out[3]))
    return out[0], out[1], out[2], out[3]


def _grad_buf(t):
    """The parameter's contiguous fp32 .grad (a view of the flat gradient buffer) if it exists, else None."""
    if isinstance(t, torch.nn.Parameter) and t.grad is not None and t.grad.is_contiguous() and t.grad.dtype == torch.float32:
        return t.grad
    return None


def bn_backward_(dxe, p, gamma, mean, rstd, relu, per_group, sync=None, beta=None, producer_bias_grad=None):
    """In place: dxe (grad w.r.t. the normalised tensor) -> grad w.r.t. the stored pre-activation p.
    Returns (dgamma[C], dbeta[C]) summed over groups -- or (None, None) after adding them straight into
    gamma.grad / beta.grad when both exist (one launch instead of autograd's two sums + two adds).
    producer_bias_grad [C] (optional): += the per-channel sum of the result, i.e. the bias gradient of the layer that
    produced p, from the values the apply pass already holds (that layer then skips its own channel-sum pass)."""
    lib = _lib.get_lib()
    N, C = p.shape[0], p.shape[1]
    P = p[0, 0].numel()
    G = N // per_group
    ws = _bn_ws(p, N, C, P, per_group)
    sums = torch.empty((G * C, 2), dtype=torch.float64, device=p.device)
    _call(p, 'vg_bn_bwd_reduce', _p(_chk(dxe)), _p(_chk(p)), N, C, P, per_group, int(relu), _p(mean), _p(rstd), _p(ws),
             _p(sums))
    count = float(per_group * P)
    local = None
    if sync is not None:
        local = sums.clone()                       # this rank's share of dgamma / dbeta (the gradient all-reduce sums them)
        sums = sync(sums)
        count = count * sync.world_size
    parts = torch.empty((2, G, C), dtype=torch.float32, device=p.device)
    if producer_bias_grad is not None:
        assert producer_bias_grad.shape == (C,) and producer_bias_grad.is_contiguous() and producer_bias_grad.dtype == torch.float32
    _call(p, 'vg_bn_bwd_apply', _p(dxe), _p(p), N, C, P, per_group, int(relu), _p(gamma), _p(mean), _p(rstd), _p(sums),
             count, _p(parts[0]), _p(parts[1]), _p(ws), _p(producer_bias_grad), 1)
    src = local if local is not None else sums      # data parallel: this rank's share (the gradient all-reduce sums them)
    gg, bg = _grad_buf(gamma), _grad_buf(beta)
    if gg is not None and bg is not None:
        _call(p, 'vg_bn_param_grad', _p(src), G, C, _p(gg), _p(bg), 1)
        return None, None
    dg = torch.empty(C, dtype=torch.float32, device=p.device); db = torch.empty_like(dg)
    _call(p, 'vg_bn_param_grad', _p(src), G, C, _p(dg), _p(db), 0)
    return dg, db


_GROUPED_OK = {}


def _grouped_wgrad_ok(p_shape, per_group):
    """Whether vg_wgrad3d_grouped has an instance for the last decoder stage at this geometry (asked once per shape)."""
    key = (tuple(p_shape), int(per_group))
    if key not in _GROUPED_OK:
        N, C, ID, IH, IW = key[0]
        d = WgradDesc(N, C, 1, ID, IH, IW, ID + 2, IH + 2, IW + 2, 3, 3, 3, 1, 0, 0, 0, 0, 0, int(per_group))
        _GROUPED_OK[key] = _lib.get_lib().dll.vg_wgrad3d_grouped_ws_bytes(ctypes.byref(d)) >= 0
    return _GROUPED_OK[key]


def bn_backward_tconv1(dy, weight, p, gamma, mean, rstd, relu, per_group, sync=None, beta=None, producer_bias_grad=None, dw_out=None):
    """Batch-norm backward fused with the data gradient of the ONE-output-channel 3x3x3 stride-1 transposed conv behind it
    (the decoder's last stage): dy [N][1][D+2][H+2][W+2], weight [C][1][3][3][3], p [N][C][D][H][W] -> dp (new tensor).
    The C-channel gradient w.r.t. the normalised tensor is recomputed from dy and never stored.
    `dw_out` (the conv weight's gradient buffer, accumulated into) selects the one-pass form: the conv's weight gradient is taken per
    batch-norm group against the NORMALISED activation (vg_wgrad3d_grouped) and both the batch-norm reductions and dW follow from it
    algebraically (vg_bn_tconv1_sums) -- the reduce pass over p and dy (0.35 ms at batch 64 / 8 covariates) disappears; the caller
    must then NOT run the layer's ordinary weight gradient.  Returns (dp, dgamma, dbeta) with the conventions of bn_backward_."""
    lib = _lib.get_lib()
    N, C = p.shape[0], p.shape[1]
    ID, IH, IW = p.shape[2:]
    assert tuple(dy.shape) == (N, 1, ID + 2, IH + 2, IW + 2) and tuple(weight.shape) == (C, 1, 3, 3, 3)
    _chk(dy); _chk(p)
    w = _chk(weight.detach().contiguous())
    G = N // per_group
    P = ID * IH * IW
    ws = torch.empty(lib.size('vg_bn_tconv1_ws_bytes', N, C, ID, per_group) // 8, dtype=torch.float64, device=p.device)
    sums = torch.empty((G * C, 2), dtype=torch.float64, device=p.device)
    if dw_out is not None:
        assert dw_out.shape == weight.shape and dw_out.is_contiguous() and dw_out.dtype == torch.float32 and beta is not None
        d = WgradDesc(N, C, 1, ID, IH, IW, ID + 2, IH + 2, IW + 2, 3, 3, 3, 1, 0, 0, 0, 0, int(relu), int(per_group))
        nbytes = lib.size('vg_wgrad3d_grouped_ws_bytes', ctypes.byref(d))
        wws = torch.empty(nbytes // 4, dtype=torch.float32, device=p.device)
        q = torch.empty((G, C + 1, 27), dtype=torch.float32, device=p.device)
        nshift = -(mean * rstd)
        _call(p, 'vg_wgrad3d_grouped', ctypes.byref(d), _p(dy), _p(p), _p(rstd), _p(nshift), _p(wws), _p(q))
        _call(p, 'vg_bn_tconv1_sums', _p(q), _p(w), _p(gamma), _p(beta), G, C, 27, _p(sums), _p(dw_out), 1)
    else:
        _call(p, 'vg_bn_bwd_reduce_tconv1', _p(dy), _p(w), _p(p), N, C, ID, IH, IW, per_group, int(relu), _p(mean), _p(rstd), _p(ws), _p(sums))
    count = float(per_group * P)
    local = None
    if sync is not None:
        local = sums.clone()
        sums = sync(sums)
        count = count * sync.world_size
    dp = torch.empty_like(p)
    if producer_bias_grad is not None:
        assert producer_bias_grad.shape == (C,) and producer_bias_grad.is_contiguous() and producer_bias_grad.dtype == torch.float32
    _call(p, 'vg_bn_bwd_apply_tconv1', _p(dy), _p(w), _p(p), _p(dp), N, C, ID, IH, IW, per_group, int(relu), _p(gamma), _p(mean), _p(rstd),
          _p(sums), count, _p(ws), _p(producer_bias_grad), 1)
    src = local if local is not None else sums
    gg, bg = _grad_buf(gamma), _grad_buf(beta)
    if gg is not None and bg is not None:
        _call(p, 'vg_bn_param_grad', _p(src), G, C, _p(gg), _p(bg), 1)
        return dp, None, None
    dg = torch.empty(C, dtype=torch.float32, device=p.device); db = torch.empty_like(dg)
    _call(p, 'vg_bn_param_grad', _p(src), G, C, _p(dg), _p(db), 0)
    return dp, dg, db


def channel_sum(x, out=None):
    lib = _lib.get_lib()
    N, C = x.shape[0], x.shape[1]
    P = x[0, 0].numel()
    ws = _bn_ws(x, N, C, P, N)
    if out is not None:          # accumulate into an existing buffer (a bias .grad)
        assert out.shape == (C,) and out.is_contiguous() and out.dtype == torch.float32
        _call(x, 'vg_channel_sum', _p(_chk(x)), N, C, P, _p(ws), _p(out), 1)
        return None
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    _call(x, 'vg_channel_sum', _p(_chk(x)), N, C, P, _p(ws), _p(out), 0)
    return out


# --------------------------------------------------------------------------- autograd nodes
class BnConvAct(torch.autograd.Function):
    """y = conv( BN( relu?(p_in) ) ) + bias, everything stored pre-activation.

    forward  : [bn_stats] -> corr3d / tconv3d_s2 with the ReLU + affine applied while staging tiles
    backward : channel_sum (bias), wgrad3d, data gradient (+ fused ReLU mask), [bn backward]
    `input_is_data=True` (first encoder layer): no data gradient is formed; the batch-norm
    parameter gradients follow from the weight gradient (valid conv, every tap hits the input).

    Two explicit hand-offs between neighbouring layers (arguments, no state outside the nodes):
      forward   `next_bn`   : this (stride-2 transposed-conv) layer also accumulates the statistics partials of the
                              BatchNorm3d that consumes its output; returned as a second output, which the caller passes
                              to that layer as `pre_stats`;
      backward  `producer_bias` : Parameter (bias of the layer that produced p_in).  This layer's batch-norm backward adds
                              the per-channel sum of the gradient it hands back straight into producer_bias.grad, and the
                              producing layer is built with `bias_grad_by_consumer=True` so that it skips its own pass.
    """

    @staticmethod
    def forward(ctx, p_in, weight, bias, gamma, beta, spec: ConvSpec, relu_in: bool, per_group: int,
                input_is_data: bool, sync, packed=None, next_bn=None, pre_stats=None, producer_bias=None,
                bias_grad_by_consumer=False):
        p_in = p_in.contiguous()
        has_bn = gamma is not None
        assert producer_bias is None or has_bn, 'producer_bias: the hand-off happens in the batch-norm backward'
        scale = shift = mean = rstd = None
        part = None
        with label(spec.name + '/fwd'):
            if has_bn:
                scale, shift, mean, rstd = bn_stats(p_in, gamma, beta, relu_in, per_group, sync, pre_stats)
            mmf = _mm_for(packed, weight, spec, 'fwd', tuple(p_in.shape[2:]), None)
            if mmf is not None:
                y = conv_mm(p_in, mmf[0], mmf[1], bias, relu_in, scale, shift, per_group, None, next_bn)
            else:
                wf = packed.get(spec.name, 'fwd') if packed is not None else pack_weight(weight, spec, 'fwd')
                y = conv_forward(p_in, wf, bias, spec, relu_in, scale, shift, per_group, next_bn)
            if next_bn:
                y, part = y
        ctx.spec, ctx.relu_in, ctx.per_group, ctx.input_is_data, ctx.sync, ctx.has_bn = \
            spec, relu_in, per_group, input_is_data, sync, has_bn
        ctx.save_for_backward(p_in, weight, gamma, beta, scale, shift, mean, rstd)
        ctx.bias_ref = bias
        ctx.packed = packed
        ctx.producer_bias, ctx.bias_grad_by_consumer = producer_bias, bool(bias_grad_by_consumer)
        ctx.nout = 2 if next_bn else 1
        if next_bn:
            ctx.mark_non_differentiable(part)
            return y, part
        return y

    @staticmethod
    def backward(ctx, dy, *_unused):
        with label(ctx.spec.name + '/bwd'):
            return BnConvAct._backward(ctx, dy) + (None, None, None)

    @staticmethod
    def _backward(ctx, dy):
        p_in, weight, gamma, beta, scale, shift, mean, rstd = ctx.saved_tensors
        spec, relu_in, per_group = ctx.spec, ctx.relu_in, ctx.per_group
        dy = dy.contiguous()
        # parameter gradients go straight into the parameters' .grad views of the flat gradient buffer when those
        # exist (no separate accumulate kernels); autograd then gets None for them
        wg = weight.grad if (isinstance(weight, torch.nn.Parameter) and weight.grad is not None and weight.grad.is_contiguous()) else None
        bias_t = ctx.bias_ref
        bg = bias_t.grad if (isinstance(bias_t, torch.nn.Parameter) and bias_t.grad is not None and bias_t.grad.is_contiguous()) else None
        direct_db = bg is not None and not (ctx.input_is_data and ctx.has_bn)
        overlap = direct_db and wg is not None and not ctx.input_is_data
        skip_db = ctx.bias_grad_by_consumer            # the consuming layer's batch-norm backward already added it to bias.grad
        # last decoder stage (one output channel, 3x3x3 stride-1 transposed conv behind a batch norm): batch-norm backward fused with
        # the recomputed data gradient; with WGRAD_BN_SUMS its reductions AND dW come out of one grouped weight-gradient launch
        fuse_last = FUSE_LAST_BN_BWD and ctx.has_bn and not ctx.input_is_data and spec.kind == 'convt' and spec.stride == 1 \
            and spec.co == 1 and tuple(spec.k) == (3, 3, 3) and tuple(spec.pad) == (0, 0, 0) and spec.ci <= 16
        dw_by_bn = fuse_last and WGRAD_BN_SUMS and _grouped_wgrad_ok(p_in.shape, per_group)
        if dw_by_bn:
            overlap = False
        if overlap:
            with on_side_stream(dy, dy, p_in, scale, shift, wg, bg):
                if not skip_db:
                    channel_sum(dy, out=bg)
                conv_weight_grad(p_in, dy, spec, relu_in, scale, shift, per_group, out=wg)
            db = dw = None
        elif skip_db:
            db = None
        else:
            db = channel_sum(dy, out=bg) if direct_db else channel_sum(dy)
        dgamma = dbeta = dp = None
        if ctx.input_is_data:
            if ctx.has_bn:
                assert spec.kind == 'conv' and spec.pad == (0, 0, 0) and not relu_in
                G = p_in.shape[0] // per_group
                assert G == 1
                # weight gradient against the NORMALISED input xhat = (x-mean)*rstd, then
                #   dw = gamma*dw_hat + beta*db ;  dgamma = <w, dw_hat> ; dbeta = <sum_k w, db>
                # (two small launches instead of 14 torch ones at the very end of the step: vg_data_bn_nshift, vg_data_bn_grads)
                nshift = torch.empty_like(mean)
                _call(mean, 'vg_data_bn_nshift', _p(_chk(mean)), _p(_chk(rstd)), mean.numel(), _p(nshift))
                dw_hat = conv_weight_grad(p_in, dy, spec, False, rstd, nshift, per_group)
                gg, gbt = _grad_buf(gamma), _grad_buf(beta)
                acc = wg is not None and bg is not None and gg is not None and gbt is not None
                taps = spec.k[0] * spec.k[1] * spec.k[2]
                dw = wg if acc else torch.empty_like(weight)
                dbo = bg if acc else torch.empty_like(db)
                dgamma = gg if acc else torch.empty_like(gamma)
                dbeta = gbt if acc else torch.empty_like(beta)
                _call(dy, 'vg_data_bn_grads', _p(_chk(dw_hat)), _p(_chk(db)), _p(_chk(weight)), _p(_chk(gamma)), _p(_chk(beta)),
                      spec.co, spec.ci, taps, _p(dw), _p(dbo), _p(dgamma), _p(dbeta), int(acc))
                if acc:
                    return None, None, None, None, None, None, None, None, None, None, None, None
                db = dbo
            else:
                dw = conv_weight_grad(p_in, dy, spec, relu_in, None, None, per_group, out=wg)
            return None, dw, db, dgamma, dbeta, None, None, None, None, None, None, None
        if dw_by_bn:
            dw = None
        elif not overlap:
            dw = conv_weight_grad(p_in, dy, spec, relu_in, scale, shift, per_group, out=wg)
        in_size = tuple(p_in.shape[2:])
        mmb = _mm_for(ctx.packed, weight, spec, 'bwd', tuple(dy.shape[2:]), in_size)
        wb = None
        if mmb is None:
            wb = ctx.packed.get(spec.name, 'bwd') if ctx.packed is not None else pack_weight(weight, spec, 'bwd')
        if ctx.has_bn:
            pbg = None
            if ctx.producer_bias is not None:
                pb = ctx.producer_bias
                if pb.grad is None:
                    # allocating here would leave the gradient OUTSIDE the flat buffer the optimiser and the all-reduce read
                    raise RuntimeError('bn_conv_act: producer_bias.grad is not bound -- the consumer adds the producer\'s bias gradient into it '
                                       '(FusedAdam binds every .grad to its flat buffer; a free-standing caller sets .grad = zeros first)')
                pbg = pb.grad
            if fuse_last:
                # the data gradient is recomputed inside the batch-norm backward pass(es), never stored
                dw_t = (wg if wg is not None else torch.zeros_like(weight)) if dw_by_bn else None
                dp, dgamma, dbeta = bn_backward_tconv1(dy, weight, p_in, gamma, mean, rstd, relu_in, per_group, ctx.sync, beta, pbg, dw_out=dw_t)
                if dw_by_bn and wg is None:
                    dw = dw_t
            else:
                dp = conv_mm(dy, mmb[0], mmb[1], None, False, None, None, 1, None) if mmb is not None else \
                    conv_backward_data(dy, wb, spec, in_size, None)
                dgamma, dbeta = bn_backward_(dp, p_in, gamma, mean, rstd, relu_in, per_group, ctx.sync, beta, pbg)
        else:
            msk = p_in if relu_in else None
            dp = conv_mm(dy, mmb[0], mmb[1], None, False, None, None, 1, msk) if mmb is not None else \
                conv_backward_data(dy, wb, spec, in_size, msk)
        return dp, dw, db, dgamma, dbeta, None, None, None, None, None, None, None


_MM_SKIP = set(filter(None, _os.environ.get('VG_MM_SKIP', '').split(',')))


def mm_wins(plan):
    """Where vg_conv_mm is the faster (or an equally fast) engine -- MI355X, tools/layer_bench.py at batch 64 / 8 covariates, us,
    register-tiled kernel -> matrix-core kernel:
      blocks that hold a whole sample: convt1 fwd 234 -> 84, bwd 87 -> 50; convt2 bwd 238 -> 120; conv5 fwd 30 -> 20;
      the stride-2 transposed conv of the decoder (convt4 fwd, 4 parity classes): 890-930 -> 797, and its output is written
      with 1.04x instead of 1.88x write traffic;
      the 3x3x3 stride-1 layers with 8 <-> 16 channels (convt3 fwd 520-534 -> 516, bwd 442-446 -> 444; conv3): ties -- taken, so
      that the matrix cores carry every launch that is a real contraction and the register-tiled kernels keep the 1-channel ends.
    Kept on the register-tiled kernels: stride-2 correlations (convt4's data gradient 574 vs 714: one input channel fills the
    LDS budget, a unit is 57 MFMAs per wave between barriers; conv2 / conv4 forward)."""
    if plan is None:
        return False
    if _MM_SKIP and 'k%d' % plan.ks[0] in _MM_SKIP and (plan.PDT + plan.PD - 1) // plan.PD > 1:     # tuning knob: VG_MM_SKIP=k7,k9 -> those go register-tiled
        return False
    if USE_MM >= 2 or (plan.PDT + plan.PD - 1) // plan.PD == 1:
        return True
    if plan.mode == 'tconv':
        return list(plan.ks) in ([3, 2, 2, 1], [2, 1, 1, 1], [2, 2, 2, 2])
    return plan.ks[0] in (7, 9)


def _mm_for(packed, weight, spec, direction, read_size, write_size):
    """(plan, A image) of the matrix-core kernel for this launch, or None (-> register-tiled kernels)."""
    if not USE_MM or (direction == 'bwd' and spec.co == 1):
        return None
    if packed is not None:
        got = packed.get_mm(spec.name, direction)
        if got is not None and (got[0].ID, got[0].IH, got[0].IW) == tuple(read_size):
            return got if mm_wins(got[0]) else None
    plan = mm_plan(spec, direction, read_size, write_size)
    if not mm_wins(plan):
        return None
    return plan, plan.gather(weight)


def bn_conv_act(p_in, weight, bias, gamma, beta, spec, relu_in, per_group=None, input_is_data=False, sync=None, packed=None,
                next_bn=None, pre_stats=None, producer_bias=None, bias_grad_by_consumer=False):
    """-> y, or (y, statistics partials for the consuming BatchNorm3d) when next_bn is given."""
    if per_group is None:
        per_group = p_in.shape[0]
    return BnConvAct.apply(p_in, weight, bias, gamma, beta, spec, relu_in, per_group, input_is_data, sync, packed, next_bn,
                           pre_stats, producer_bias, bias_grad_by_consumer)


class GamElbo(torch.autograd.Function):
    """(logits[G,B,V], gain[C,B], x[B,V], eps[V] f64, glm[C,V]) -> (sum_log_prob[B], dist[C,B]).
    `logits_bias` (optional Parameter of one element): the bias of the one-output-channel layer that produced `logits`; the backward
    then adds sum(d_logits) into its .grad (explicit hand-off: that layer is built with bias_grad_by_consumer=True)."""

    @staticmethod
    def forward(ctx, logits, gain, x, eps, glm, logits_bias=None):
        lib = _lib.get_lib()
        G, B, V = logits.shape
        C = G - 1
        logits = _chk(logits.contiguous()); gain = _chk(gain.contiguous()); x = _chk(x.contiguous())
        eps = _chk(eps.contiguous(), torch.float64); glm = _chk(glm.contiguous())
        ws = torch.empty(lib.size('vg_gam_ws_bytes', C, B, V) // 4 + 1, dtype=torch.float32, device=x.device)
        slp = torch.empty(B, dtype=torch.float32, device=x.device)
        dist = torch.empty((C, B), dtype=torch.float32, device=x.device)
        _call(x, 'vg_gam_elbo_fwd', _p(logits), _p(gain), _p(x), _p(eps), _p(glm), C, B, V, _p(ws), _p(slp), _p(dist),
                 None)
        ctx.save_for_backward(logits, gain, x, eps, glm, dist)
        ctx.logits_bias = logits_bias
        return slp, dist

    @staticmethod
    def backward(ctx, g_slp, g_dist):
        lib = _lib.get_lib()
        logits, gain, x, eps, glm, dist = ctx.saved_tensors
        tot = None
        if ctx.logits_bias is not None:
            pb = ctx.logits_bias
            assert pb.numel() == 1
            if pb.grad is None:
                raise RuntimeError('GamElbo: logits_bias.grad is not bound (see bn_conv_act: the bias gradient is added into the caller\'s buffer)')
            tot = pb.grad
        G, B, V = logits.shape
        C = G - 1
        g_slp = _chk(g_slp.contiguous()); g_dist = _chk(g_dist.contiguous())
        ws = torch.empty(lib.size('vg_gam_ws_bytes', C, B, V) // 4 + 1, dtype=torch.float32, device=x.device)
        d_logits = torch.empty_like(logits)
        d_gain = torch.empty_like(gain)
        d_eps = torch.empty_like(eps)
        _call(x, 'vg_gam_elbo_bwd', _p(logits), _p(gain), _p(x), _p(eps), _p(glm), _p(dist), _p(g_slp), _p(g_dist),
                 C, B, V, _p(ws), _p(d_logits), _p(d_gain), _p(d_eps), _p(tot), 1)
        return d_logits, d_gain, None, d_eps, None, None


class LatentSample(torch.autograd.Function):
    """(mu, w, a, eps_w, eps_d, G) -> (zcat[G*B, L+G], kl_z[B], d[B, L]); one launch each way
    (vae_reg_GP.py:321-329, 339-342, 400).  `a` is fc43's output (d = exp(a) + the batch-wide 1e-6 floor)."""

    @staticmethod
    def forward(ctx, mu, w, a, eps_w, eps_d, G):
        B, L = mu.shape
        mu = _chk(mu.contiguous()); w = _chk(w.contiguous()); a = _chk(a.contiguous())
        eps_w = _chk(eps_w.contiguous()); eps_d = _chk(eps_d.contiguous())
        assert w.shape == (B, L) and a.shape == (B, L) and eps_d.shape == (B, L) and eps_w.numel() == B
        zcat = torch.empty((G * B, L + G), dtype=torch.float32, device=mu.device)
        kl = torch.empty(B, dtype=torch.float32, device=mu.device)
        d = torch.empty((B, L), dtype=torch.float32, device=mu.device)
        flag = torch.empty(1, dtype=torch.float32, device=mu.device)
        _call(mu, 'vg_latent_fwd', _p(mu), _p(w), _p(a), _p(eps_w), _p(eps_d), B, L, G, _p(zcat), _p(kl), _p(d), _p(flag))
        ctx.save_for_backward(mu, w, d, flag, eps_w, eps_d)
        ctx.G = G
        ctx.mark_non_differentiable(d)
        return zcat, kl, d

    @staticmethod
    def backward(ctx, g_zcat, g_kl, _g_d):
        mu, w, d, flag, eps_w, eps_d = ctx.saved_tensors
        B, L = mu.shape
        g_mu = torch.empty_like(mu); g_w = torch.empty_like(mu); g_a = torch.empty_like(mu)
        gz = _p(_chk(g_zcat.contiguous())) if g_zcat is not None else None
        gk = _p(_chk(g_kl.contiguous())) if g_kl is not None else None
        _call(mu, 'vg_latent_bwd', _p(mu), _p(w), _p(d), _p(flag), _p(eps_w), _p(eps_d), gz, gk, B, L, ctx.G,
                 _p(g_mu), _p(g_w), _p(g_a))
        return g_mu, g_w, g_a, None, None, None


class LatentSampleStacked(torch.autograd.Function):
    """LatentSample on the stacked heads output mwa[3, B, L] = (mu, w, a): one gradient tensor goes back to HeadsAct
    (indexing mwa[0..2] under autograd costs three select-backward fills + copies + two adds per step)."""

    @staticmethod
    def forward(ctx, mwa, eps_w, eps_d, G):
        mwa = _chk(mwa.contiguous())
        _, B, L = mwa.shape
        eps_w = _chk(eps_w.contiguous()); eps_d = _chk(eps_d.contiguous())
        assert eps_d.shape == (B, L) and eps_w.numel() == B
        zcat = torch.empty((G * B, L + G), dtype=torch.float32, device=mwa.device)
        kl = torch.empty(B, dtype=torch.float32, device=mwa.device)
        d = torch.empty((B, L), dtype=torch.float32, device=mwa.device)
        flag = torch.empty(1, dtype=torch.float32, device=mwa.device)
        _call(mwa, 'vg_latent_fwd', _p(mwa[0]), _p(mwa[1]), _p(mwa[2]), _p(eps_w), _p(eps_d), B, L, G, _p(zcat), _p(kl), _p(d), _p(flag))
        ctx.save_for_backward(mwa, d, flag, eps_w, eps_d)
        ctx.G = G
        ctx.mark_non_differentiable(d)
        return zcat, kl, d

    @staticmethod
    def backward(ctx, g_zcat, g_kl, _g_d):
        mwa, d, flag, eps_w, eps_d = ctx.saved_tensors
        _, B, L = mwa.shape
        g = torch.empty_like(mwa)
        gz = _p(_chk(g_zcat.contiguous())) if g_zcat is not None else None
        gk = _p(_chk(g_kl.contiguous())) if g_kl is not None else None
        _call(mwa, 'vg_latent_bwd', _p(mwa[0]), _p(mwa[1]), _p(d), _p(flag), _p(eps_w), _p(eps_d), gz, gk, B, L, ctx.G,
                 _p(g[0]), _p(g[1]), _p(g[2]))
        return g, None, None, None


class ElboLoss(torch.autograd.Function):
    """loss[1] = c_kl*sum(kl_z) + c_slp*sum(slp) + c_gp*gp_kl + c_dist*sum(dist)   (vae_reg_GP.py:406-410)."""

    @staticmethod
    def forward(ctx, kl_z, slp, dist, gp_kl, coef):
        kl_z = _chk(kl_z.contiguous()); slp = _chk(slp.contiguous()); dist = _chk(dist.contiguous()); gp_kl = _chk(gp_kl.contiguous())
        loss = torch.empty(1, dtype=torch.float32, device=kl_z.device)
        ctx.coef = tuple(float(c) for c in coef)
        ctx.shapes = (kl_z.shape, slp.shape, dist.shape, gp_kl.shape)
        _call(kl_z, 'vg_loss_fwd', _p(kl_z), _p(slp), _p(dist), _p(gp_kl), kl_z.numel(), dist.numel(), *ctx.coef, _p(loss))
        return loss

    @staticmethod
    def backward(ctx, g):
        sk, ss, sd, sg = ctx.shapes
        g = _chk(g.contiguous())
        g_kl = torch.empty(sk, dtype=torch.float32, device=g.device); g_slp = torch.empty(ss, dtype=torch.float32, device=g.device)
        g_dist = torch.empty(sd, dtype=torch.float32, device=g.device); g_gp = torch.empty(sg, dtype=torch.float32, device=g.device)
        _call(g, 'vg_loss_bwd', _p(g), g_kl.numel(), g_dist.numel(), *ctx.coef, _p(g_kl), _p(g_slp), _p(g_dist), _p(g_gp))
        return g_kl, g_slp, g_dist, g_gp, None


def _fc_split(M, N, K):
    """Split-K factor of a fully connected product.  The kernel keeps four 64-index steps in flight, so a reduction of a few hundred
    indices is one or two memory round trips and is left whole; a LONG reduction onto few output tiles (fc1: 3072 -> 64 x 200; fc8's data
    gradient: 3840 -> 576 x 200) is cut into pieces of >= 256 indices until ~1024 blocks exist.  The pieces are sized as the kernel
    sizes them (ceil(K / ksplit) rounded up to 64) and none may be empty."""
    tiles = ((M + 31) // 32) * ((N + 31) // 32)
    if K <= 1024:
        return 1
    ks = max(1, min(K // 256, 1024 // tiles))
    if ks <= 1:
        return 1
    kchunk = (((K + ks - 1) // ks) + 63) // 64 * 64
    return (K + kchunk - 1) // kchunk


def fc_job(A, B, C, M, N, K, a_strides, b_strides, c_strides, flags=0, batch=1, bias=None, bias_sb=0, amask=None, cmask=None,
           cx=None, cx_sb=0, ksplit=None):
    """One product for vg_fc_gemm_jobs: C[z][m][n] (+)= epilogue(sum_k A[z](m,k) B[z](k,n)); a_strides = (sm, sk, sb),
    b_strides = (sk, sn, sb), c_strides = (sm, sb), all in elements (include/vaegam.h).  Returns (job struct, tensors to keep alive)."""
    lib = _lib.get_lib()
    if ksplit is None:
        ksplit = _fc_split(M, N + (1 if flags & _lib.FC_B_ONES else 0), K) if batch == 1 else 1
    d = _lib.FcDesc(M, N, K, batch, a_strides[0], a_strides[1], a_strides[2], b_strides[0], b_strides[1], b_strides[2],
                    c_strides[0], c_strides[1], bias_sb, cx_sb, ksplit, flags)
    ws = None
    if ksplit > 1:
        ws = torch.empty(lib.size('vg_fc_ws_bytes', ctypes.byref(d)) // 4, dtype=torch.float32, device=C.device)
    ptr = lambda t: None if t is None else t.data_ptr()
    return _lib.FcJob(d, ptr(A), ptr(amask), ptr(B), ptr(bias), ptr(cmask), ptr(C), ptr(cx), ptr(ws)), (A, B, C, bias, amask, cmask, cx, ws)


def fc_launch(ref, jobs):
    """vg_fc_gemm_jobs: the products of `jobs` (fc_job results, <= 4) in one launch on ref's stream."""
    arr = (_lib.FcJob * len(jobs))(*[j[0] for j in jobs])
    _call(ref, 'vg_fc_gemm_jobs', arr, len(jobs))


def fc_gemm(A, B, C, *args, **kw):
    """A single product (see fc_job)."""
    fc_launch(C, [fc_job(A, B, C, *args, **kw)])
    return C


def _fc_forward(x, weight, bias, relu, relu_in):
    M, K = x.shape; N = weight.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    fl = _lib.FC_C_BIAS | (_lib.FC_C_RELU if relu else 0) | (_lib.FC_A_RELU if relu_in else 0)
    return fc_gemm(x, weight, y, M, N, K, (x.stride(0), 1, 0), (1, weight.stride(0), 0), (N, 0), fl, bias=bias)


def _fc_backward(gy, x, weight, y, relu_in, need_gx, wg, bg):
    """Gradients of y = [relu](x' W^T + b), x' = relu(x) if relu_in: gy is masked with y > 0 inside both products (y None: no ReLU);
    -> (gx or None, gw or None, gb or None); wg / bg (the .grad views) are added into when given.  ONE launch: the data gradient and
    the weight gradient (whose extra column is the bias gradient) are two jobs of vg_fc_gemm_jobs."""
    M, N = gy.shape; K = x.shape[1]
    am = _lib.FC_A_MASK if y is not None else 0
    jobs = []
    gx = None
    if need_gx:
        gx = torch.empty((M, K), dtype=torch.float32, device=gy.device)
        jobs.append(fc_job(gy, weight, gx, M, K, N, (N, 1, 0), (weight.stride(0), 1, 0), (K, 0), am | (_lib.FC_C_MASK if relu_in else 0),
                           amask=y, cmask=x if relu_in else None))
    acc = wg is not None and bg is not None
    gw = wg if acc else torch.empty((N, K), dtype=torch.float32, device=gy.device)
    gb = bg if acc else torch.empty((N,), dtype=torch.float32, device=gy.device)
    jobs.append(fc_job(gy, x, gw, N, K, M, (1, N, 0), (x.stride(0), 1, 0), (K, 0),
                       am | _lib.FC_B_ONES | (_lib.FC_B_RELU if relu_in else 0) | (_lib.FC_C_ACCUM if acc else 0), amask=y, cx=gb))
    fc_launch(gy, jobs)
    return gx, (None if acc else gw), (None if acc else gb)


class LinearAct(torch.autograd.Function):
    """y = [relu]([relu](x) @ W^T + b) for the fully connected layers (vae_reg_GP.py:203-209, 224-234, 243-259) on vg_fc_gemm: bias and
    ReLU in the product's epilogue, the ReLU of a pre-activation input (conv5's output, :243) in its operand load; the backward is two
    launches -- the data gradient and the weight gradient whose extra column is the bias gradient, both masking the incoming gradient
    with y > 0 on load and adding dW / db straight into the .grad views of the flat gradient buffer."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, relu_in=False):
        x = _chk(x.contiguous()); _chk(weight); _chk(bias)
        y = _fc_forward(x, weight, bias, relu, relu_in)
        ctx.relu, ctx.relu_in = relu, relu_in
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.bias_ref = bias
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        bias = ctx.bias_ref
        gy = _chk(gy.contiguous())
        gx, gw, gb = _fc_backward(gy, x, weight, y, ctx.relu_in, ctx.needs_input_grad[0], _grad_buf(weight), _grad_buf(bias))
        return gx, gw, gb, None, None


class HeadsAct(torch.autograd.Function):
    """The three encoder heads (vae_reg_GP.py:205-209, 246-252: fc31/32/33 -> ReLU -> fc41/42/43) as ONE product over the stacked
    first-stage weights and ONE batched product over the second stage.  W3 [3*H, F], b3 [3*H], W4 [3, L, H], b4 [3, 1, L] are
    views of the flat parameter buffer (the optimiser lays the six weights / six biases out back to back), gW3.. the matching
    views of the flat gradient buffer, which the backward adds into directly.  9 -> 2 launches forward, 23 -> 2 backward."""

    @staticmethod
    def forward(ctx, h, W3, b3, W4, b4, gW3, gb3, gW4, gb4):
        B = h.shape[0]
        h = _chk(h.contiguous())
        G, L, H = W4.shape
        y = _fc_forward(h, W3, b3, True, False)                          # (B, 3H)
        out = torch.empty((G, B, L), dtype=torch.float32, device=h.device)
        fc_gemm(y, W4, out, B, L, H, (G * H, 1, H), (1, H, L * H), (L, B * L), _lib.FC_C_BIAS, batch=G, bias=b4, bias_sb=L)
        ctx.save_for_backward(h, y, W3, W4)
        ctx.grads = (gW3, gb3, gW4, gb4)
        return out

    @staticmethod
    def backward(ctx, gout):
        h, y, W3, W4 = ctx.saved_tensors
        gW3, gb3, gW4, gb4 = ctx.grads
        B = h.shape[0]
        G, L, H = W4.shape
        gout = _chk(gout.contiguous())
        gX = torch.empty((B, G * H), dtype=torch.float32, device=h.device)           # gradient w.r.t. y, before its ReLU mask
        fc_launch(gout, [fc_job(gout, W4, gX, B, H, L, (L, 1, B * L), (H, 1, L * H), (G * H, H), 0, batch=G),
                         fc_job(gout, y, gW4, L, H, B, (1, L, B * L), (G * H, 1, H), (H, L * H), _lib.FC_B_ONES | _lib.FC_C_ACCUM, batch=G,
                                cx=gb4, cx_sb=L)])
        gh, _, _ = _fc_backward(gX, h, W3, y, False, True, gW3, gb3)
        return gh, None, None, None, None, None, None, None, None


_ONES = {}


def _ones(n, device):
    key = (n, str(device))
    if key not in _ONES:
        _ONES[key] = torch.ones(n, dtype=torch.float32, device=device)
    return _ONES[key]


def linear_act(layer, x, relu, relu_in=False):
    return LinearAct.apply(x, layer.weight, layer.bias, relu, relu_in)


def gam_maps(logits, gain, x, eps, glm):
    """Reconstruction path (vae_reg_GP.py:331,391-392): the C+2 maps [base, cons_1..C, full_rec] as [C+2,B,V]."""
    lib = _lib.get_lib()
    G, B, V = logits.shape
    C = G - 1
    ws = torch.empty(lib.size('vg_gam_ws_bytes', C, B, V) // 4 + 1, dtype=torch.float32, device=x.device)
    slp = torch.empty(B, dtype=torch.float32, device=x.device)
    dist = torch.empty((max(C, 1), B), dtype=torch.float32, device=x.device)
    maps = torch.empty((G + 1, B, V), dtype=torch.float32, device=x.device)
    _call(x, 'vg_gam_elbo_fwd', _p(_chk(logits.contiguous())), _p(_chk(gain.contiguous())), _p(_chk(x.contiguous())),
             _p(_chk(eps.contiguous(), torch.float64)), _p(_chk(glm.contiguous())), C, B, V, _p(ws), _p(slp), _p(dist),
             _p(maps))
    return maps


class CholeskyF64(torch.autograd.Function):
    """L = chol(A) for a batch of small float64 SPD matrices (vg_cholesky_f64; n <= 128).  Backward is the
    standard  dA = sym( L^-T  Phi(L^T dL)  L^-1 )  with two triangular solves (rocBLAS trsm, capturable)."""

    @staticmethod
    def forward(ctx, a):
        assert a.dtype == torch.float64 and a.dim() == 3 and a.shape[1] == a.shape[2]
        a = a.contiguous()
        l = torch.empty_like(a)
        _call(a, 'vg_cholesky_f64', _p(a), _p(l), a.shape[0], a.shape[1])
        ctx.save_for_backward(l)
        return l

    @staticmethod
    def backward(ctx, gl):
        (l,) = ctx.saved_tensors
        p = (l.transpose(1, 2) @ gl).tril()
        p = p - 0.5 * torch.diag_embed(p.diagonal(dim1=1, dim2=2))
        x = torch.linalg.solve_triangular(l.transpose(1, 2), p, upper=True)              # L^-T P
        s = torch.linalg.solve_triangular(l, x, upper=False, left=False)                  # (L^-T P) L^-1
        return 0.5 * (s + s.transpose(1, 2))


def cholesky(a):
    """Batched lower Cholesky factor; the HIP kernel for float64 n <= 128, torch otherwise."""
    if a.dtype == torch.float64 and a.shape[-1] <= 128 and (a.is_cuda or _lib.get_lib().host_pointers_ok):
        return CholeskyF64.apply(a)
    return torch.linalg.cholesky_ex(a, check_errors=False).L


class GainConsts:
    """Device-side constants of the gain block of one model: the parameter-offset table, the inducing grids, the HRF taps."""

    def __init__(self, table, xu, hrf, n, jitter_ku=0.0, jitter_b=1e-5, prior_var=10.0):
        self.table, self.xu, self.hrf = table, xu, hrf            # int64 [C][10], fp32 [K][n] (or None), float64 [taps]
        self.C, self.n = int(table.shape[0]), int(n)
        self.jitter_ku, self.jitter_b, self.prior_var = float(jitter_ku), float(jitter_b), float(prior_var)

    def desc(self, B):
        return _lib.GainDesc(self.C, int(B), self.n, int(self.hrf.numel()), self.jitter_b, self.jitter_ku, self.prior_var)


class GpGain(torch.autograd.Function):
    """(covariates (B, >=C) fp32, eps_beta (C, B) fp32) -> gains task_var (C, B) fp32, gp_kl (1,) fp32 and float64 copies of
    beta_mean (C,B), beta_cov (C,B,B), f_bar (C,B), Sigma (C,B,B), kl terms (C,) for exports / tests: vg_gp_gain_fwd, one
    workgroup per covariate (vae_reg_GP.py:345-378, gp.py:41-110).  The backward launch adds the gain-parameter gradients
    straight into the flat fp32 gradient buffer; `params` are listed only so that autograd calls it.
    `join_stream`: the stream that must wait for the backward launch (the model runs this block on a second stream)."""

    @staticmethod
    def forward(ctx, covariates, eps_beta, consts, flat_p, flat_g, join_stream, *params):
        lib = _lib.get_lib()
        B = covariates.shape[0]
        C = consts.C
        assert covariates.dtype == torch.float32 and covariates.dim() == 2 and covariates.shape[1] >= C and covariates.stride(1) == 1
        eps_beta = _chk(eps_beta.contiguous())
        assert eps_beta.shape == (C, B)
        dev = covariates.device
        d = consts.desc(B)
        ws = torch.empty(lib.size('vg_gp_gain_ws_bytes', C, B, consts.n) // 8, dtype=torch.float64, device=dev)
        tv = torch.empty((C, B), dtype=torch.float32, device=dev)
        kl = torch.empty(1, dtype=torch.float32, device=dev)
        bm = torch.empty((C, B), dtype=torch.float64, device=dev)
        bc = torch.empty((C, B, B), dtype=torch.float64, device=dev)
        fb = torch.zeros((C, B), dtype=torch.float64, device=dev)
        sg = torch.zeros((C, B, B), dtype=torch.float64, device=dev)
        _call(covariates, 'vg_gp_gain_fwd', ctypes.byref(d), _p(consts.table), _p(flat_p), _p(consts.xu), _p(covariates),
              int(covariates.stride(0)), _p(eps_beta), _p(consts.hrf), _p(ws), _p(tv), _p(kl), _p(bm), _p(bc), _p(fb), _p(sg))
        kl_terms = ws[-(C + 8):-8]                                # per-covariate kl_lin (+ kl_gp), float64
        ctx.save_for_backward(covariates, eps_beta, ws)
        ctx.consts, ctx.flat_p, ctx.flat_g, ctx.join_stream = consts, flat_p, flat_g, join_stream
        ctx.mark_non_differentiable(bm, bc, fb, sg, kl_terms)
        return tv, kl, bm, bc, fb, sg, kl_terms

    @staticmethod
    def backward(ctx, g_tv, g_kl, *_):
        covariates, eps_beta, ws = ctx.saved_tensors
        consts = ctx.consts
        B = covariates.shape[0]
        d = consts.desc(B)
        if g_tv is None:
            g_tv = torch.zeros((consts.C, B), dtype=torch.float32, device=covariates.device)
        if g_kl is None:
            g_kl = torch.zeros(1, dtype=torch.float32, device=covariates.device)
        g_tv = _chk(g_tv.contiguous()); g_kl = _chk(g_kl.contiguous())
        _call(covariates, 'vg_gp_gain_bwd', ctypes.byref(d), _p(consts.table), _p(ctx.flat_p), _p(consts.xu), _p(covariates),
              int(covariates.stride(0)), _p(eps_beta), _p(consts.hrf), _p(ws), _p(g_tv), _p(g_kl), _p(ctx.flat_g))
        if ctx.join_stream is not None and covariates.is_cuda:
            cur = torch.cuda.current_stream(covariates.device)
            if cur != ctx.join_stream:
                ctx.join_stream.wait_stream(cur)               # the gradients are written outside autograd's own stream bookkeeping
        return (None,) * len(ctx.needs_input_grad)


class HrfAcrossRanks(torch.autograd.Function):
    """Data parallel, dp_gain='local': the causal HRF convolution along the GLOBAL batch index (vae_reg_GP.py:283-305, 377-378) of
    gains that every rank drew for its own slice.  Forward: all-gather the pre-HRF gains of the HRF covariates ((Ch, B) -> (Ch, Bg):
    a few hundred floats), multiply by the (Bg, Bg) Toeplitz matrix of the 15 taps, keep this rank's columns.  Backward: a rank's
    loss also depends on the up to 14 volumes in front of its slice, which another rank drew -- the gradient with respect to the
    gathered gains is summed over ranks (one tiny all-reduce) and each rank keeps the columns it drew.  Without this the
    convolution would restart at every slice boundary."""

    @staticmethod
    def forward(ctx, pre, T, dp, lo):
        B = pre.shape[1]
        full = dp.all_gather_rows(pre.t().contiguous()).t()             # (Ch, Bg), rank order = batch order
        ctx.save_for_backward(T)
        ctx.dp, ctx.lo, ctx.B = dp, int(lo), int(B)
        return (full @ T)[:, lo:lo + B].contiguous()

    @staticmethod
    def backward(ctx, g):
        (T,) = ctx.saved_tensors
        lo, B = ctx.lo, ctx.B
        gp = g.contiguous() @ T[:, lo:lo + B].t()                       # (Ch, Bg): d(this rank's loss) / d(every pre-HRF gain it used)
        gp = ctx.dp.allreduce_sum_(gp)
        return gp[:, lo:lo + B].contiguous(), None, None, None


def adam_advance_(state, lr, b1, b2):
    """state = double[3] {lr/(1-b1^t), sqrt(1-b2^t), t} on the device: t += 1, scalars refreshed (one thread)."""
    _call(state, 'vg_adam_advance', _p(_chk(state, torch.float64)), float(lr), float(b1), float(b2))


def adam_step_(p, g, m, v, b1, b2, eps, step_scalars):
    """In-place fused Adam over one flat buffer (fp32 or fp64)."""
    lib = _lib.get_lib()
    assert p.dtype == g.dtype == m.dtype == v.dtype and p.dtype in (torch.float32, torch.float64)
    assert p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()
    _call(p, 'vg_adam_step', _p(p), _p(g), _p(m), _p(v), p.numel(), int(p.dtype == torch.float64), float(b1), float(b2),
             float(eps), _p(_chk(step_scalars, torch.float64)))


# --------------------------------------------------------------------------- matrix-core convolution plans (vg_conv_mm)
USE_MM = int(_os.environ.get('VG_CONV_MM', '1'))            # 0: off; 1: where it measured faster (mm_wins); 2: wherever a plan exists
_MM_LDS_BUDGET = 150 * 1024
_MM_LDS_CU = 160 * 1024                                     # LDS of a CU: what co-resident blocks share
# (mode, stride, CI, CO, kernel, read size, write size) -> (waves, PD, PHB, cc, dbuf): tools/diag/mm_sweep.py at batch 64, 8 covariates
_MM_TUNED = {
    ('corr', 1, 16, 16, (3, 3, 3), (6, 8, 5), (8, 10, 7)): (8, 4, 10, 16, 0),    # convt1 forward 66.7 us (score's choice: 78)
    ('corr', 1, 16, 16, (3, 3, 3), (8, 10, 7), (6, 8, 5)): (8, 3, 8, 16, 0),     # convt1 data gradient 37.8
    ('corr', 2, 16, 16, (3, 3, 3), (16, 21, 14), (8, 10, 7)): (8, 2, 10, 8, 0),   # convt2 data gradient 83.2
    ('corr', 1, 16, 8, (3, 3, 3), (17, 21, 14), (19, 23, 16)): (8, 1, 23, 8, 0),  # conv3 data gradient 61.6
    ('corr', 1, 8, 16, (3, 3, 3), (19, 23, 16), (17, 21, 14)): (8, 1, 21, 8, 0),  # conv3 forward 46 (score's choice, a 6-row slab: 51)
    ('corr', 1, 8, 16, (3, 3, 3), (18, 23, 16), (16, 21, 14)): (8, 2, 11, 8, 1),  # convt3 data gradient 242 (whole planes: 254)
    ('tconv', 2, 8, 8, (5, 3, 3), (18, 23, 16), (39, 47, 33)): (8, 2, 12, 8, 1),   # convt4 forward 610 (whole planes, single buffer: 656 in the same run)
}
_MM_WAVES = 8


class MmPlan:
    """Host-side description of one conv / transposed-conv launch for vg_conv_mm (include/vaegam.h): position grid, staging
    geometry and the window-offset tables, plus `aidx` -- for every A-image slot the index of the weight it holds inside the
    layer's own weight tensor (-1 = zero)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)
        self._dev = {}

    def tables(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (torch.from_numpy(self.tau).to(device), torch.from_numpy(self.dlt).to(device), torch.from_numpy(self.aidx).to(device))
        return self._dev[key]

    def desc(self, N, relu_in, per_group):
        d = _lib.MmDesc()
        for k in ('CI', 'CO', 'ID', 'IH', 'IW', 'OD', 'OH', 'OW', 'nq', 'PDT', 'PH', 'PW', 'PD', 'sdi', 'shi', 'swi', 'd0', 'LD', 'cc', 'sdo', 'sho',
                  'swo', 'tpc', 'slack', 'dbuf', 'PHB', 'hlo', 'hhi', 'waves'):
            setattr(d, k, int(getattr(self, k)))
        d.N, d.relu_in, d.per_group = int(N), int(bool(relu_in)), int(per_group)
        for q in range(4):
            d.ks[q] = int(self.ks[q]) if q < self.nq else 0
            d.od0[q] = int(self.od0[q]) if q < self.nq else 0
            d.oh0[q] = int(self.oh0[q]) if q < self.nq else 0
            d.ow0[q] = int(self.ow0[q]) if q < self.nq else 0
        return d

    def gather(self, weight):
        """A image from a weight tensor (layer tests / un-packed path; the model gathers all layers in one launch)."""
        _, _, aidx = self.tables(weight.device)
        flat = weight.detach().reshape(-1)
        a = flat[aidx.clamp_min(0).long()]
        return torch.where(aidx >= 0, a, torch.zeros_like(a)).contiguous()


def _mm_problem(spec: ConvSpec, direction: str, in_size):
    """-> (mode, S, lead_pad, CI_l, CO_l, K, out_size, widx) of the launch: mode 'corr' (strided correlation with leading zero
    padding) or 'tconv' (stride-2 transposed convolution, gather form); widx(cin, kd, kh, kw, cout) = flat index of that weight in
    the layer's own weight tensor (conv: [co][ci][k], convt: [ci][co][k])."""
    import numpy as np
    KD, KH, KW = spec.k
    kvol = KD * KH * KW
    conv = spec.kind == 'conv'

    def w_at(ci_layer, co_layer, kd, kh, kw):
        t = (kd * KH + kh) * KW + kw
        return ((co_layer * spec.ci + ci_layer) if conv else (ci_layer * spec.co + co_layer)) * kvol + t
    fl = lambda kd, kh, kw: (KD - 1 - kd, KH - 1 - kh, KW - 1 - kw)
    if direction == 'fwd':
        osz = spec.out_size(in_size)
        if conv:
            return 'corr', spec.stride, (0, 0, 0), spec.ci, spec.co, spec.k, osz, (lambda cin, kd, kh, kw, cout: w_at(cin, cout, kd, kh, kw))
        if spec.stride == 1:
            pad = tuple(spec.k[a] - 1 - spec.pad[a] for a in range(3))
            return 'corr', 1, pad, spec.ci, spec.co, spec.k, osz, (lambda cin, kd, kh, kw, cout: w_at(cin, cout, *fl(kd, kh, kw)))
        return 'tconv', 2, tuple(spec.pad), spec.ci, spec.co, spec.k, osz, (lambda cin, kd, kh, kw, cout: w_at(cin, cout, kd, kh, kw))
    # data gradient: in_size is the size of dy (the layer's OUTPUT); the launch writes the layer's input size
    if conv:
        isz = None                                                       # caller passes the target size
        if spec.stride == 1:
            pad = tuple(spec.k[a] - 1 for a in range(3))
            return 'corr', 1, pad, spec.co, spec.ci, spec.k, isz, (lambda cin, kd, kh, kw, cout: w_at(cout, cin, *fl(kd, kh, kw)))
        return 'tconv', 2, (0, 0, 0), spec.co, spec.ci, spec.k, isz, (lambda cin, kd, kh, kw, cout: w_at(cout, cin, kd, kh, kw))
    return 'corr', spec.stride, tuple(spec.pad), spec.co, spec.ci, spec.k, None, (lambda cin, kd, kh, kw, cout: w_at(cout, cin, kd, kh, kw))


_MM_PLANS = {}


def mm_plan(spec: ConvSpec, direction: str, in_size, out_size=None, force=None) -> Optional[MmPlan]:
    """Plan for vg_conv_mm, or None when the launch is not covered (1-channel ends, 16-channel stride-2 transposed convs, shapes
    that do not fit LDS): the caller then uses the register-tiled kernels.  `in_size`: spatial size of the tensor the launch READS
    (the layer input for 'fwd', dy for 'bwd'); `out_size`: spatial size it writes (needed for 'bwd')."""
    import numpy as np
    key = (spec, direction, tuple(in_size), None if out_size is None else tuple(out_size), force)
    if key in _MM_PLANS:
        return _MM_PLANS[key]
    mode, S, pad, CI, CO, K, osz, widx = _mm_problem(spec, direction, tuple(in_size))
    if osz is None:
        osz = tuple(out_size)
    plan = None
    if CO in (8, 16) and CI >= 8 and not (mode == 'tconv' and CO != 8):
        plan = _mm_build(mode, S, pad, CI, CO, K, tuple(in_size), tuple(osz), widx, force)
    _MM_PLANS[key] = plan
    return plan


def _mm_build(mode, S, pad, CI, CO, K, isz, osz, widx, force=None):
    """force = (PD, cc): tuning sweeps (tools/diag) pin the tile instead of taking the scored choice."""
    import numpy as np
    KD, KH, KW = K
    ID, IH, IW = isz
    OD, OH, OW = osz
    IHW = IH * IW
    NR = 16 // CO
    classes = []                                   # per class: list of (dd, dh, dw, tap(rho) -> (kd, kh, kw) or None)
    if mode == 'corr':
        KWp = KW + (NR - 1) * S
        ent = []
        for kd in range(KD):
            for kh in range(KH):
                for kwp in range(KWp):
                    def tap(rho, kd=kd, kh=kh, kwp=kwp):
                        kw = kwp - rho * S
                        return (kd, kh, kw) if 0 <= kw < KW else None
                    ent.append((kd - pad[0], kh - pad[1], kwp - pad[2], tap))
        classes.append(ent)
        PDT, PH, PW = OD, OH, (OW + NR - 1) // NR
        sdi, shi, swi = S, S, NR * S
        sdo, sho, swo = 1, 1, NR
        od0, oh0, ow0 = [0], [0], [0]
        d0 = -pad[0]
        ld_of = lambda PD: (PD - 1) * S + KD
    else:
        assert NR == 2 and S == 2
        MDm = (KD + 1) // 2
        od0, oh0, ow0 = [], [], []
        for rd, rh in ((0, 0), (0, 1), (1, 0), (1, 1)):
            if True:
                ent = []
                for md in range((KD - rd + 1) // 2):
                    for mh in range((KH - rh + 1) // 2):
                        for mw in range(2):
                            def tap(rho, rd=rd, rh=rh, md=md, mh=mh, mw=mw):
                                kw = rho + 2 * mw
                                return (rd + 2 * md, rh + 2 * mh, kw) if kw < KW else None
                            ent.append((-md, -mh, -mw, tap))
                classes.append(ent)
                od0.append(rd - pad[0]); oh0.append(rh - pad[1]); ow0.append(-pad[2])
        PDT, PH, PW = (OD + pad[0] + 1) // 2, (OH + pad[1] + 1) // 2, (OW + pad[2] + 1) // 2
        sdi = shi = swi = 1
        sdo = sho = swo = 2
        d0 = -(MDm - 1)
        ld_of = lambda PD: PD + MDm - 1
    nq = len(classes)
    if nq not in (1, 4) or any(len(e) == 0 for e in classes):
        return None
    ks = [(len(e) + 3) // 4 for e in classes]
    if max(ks) > 32:
        return None
    rows = sum(ks)
    tau = np.zeros((rows, 4), np.int32); dlt = np.zeros((rows, 4, 3), np.int32)
    aidx = []
    slack = 0
    r0 = 0
    for q, ent in enumerate(classes):
        a = -np.ones((CI, ks[q], 64), np.int32)
        for s in range(ks[q]):
            for kk in range(4):
                kap = 4 * s + kk
                e = ent[kap] if kap < len(ent) else ent[0]               # padding entries: any readable offset, zero weights
                dd, dh, dw = e[0], e[1], e[2]
                tau[r0 + s, kk] = dd * IHW + dh * IW + dw
                dlt[r0 + s, kk] = (dd, dh, dw)
                slack = max(slack, abs(dh) * IW + abs(dw) + 1)
                if kap >= len(ent):
                    continue
                for row in range(16):
                    rho, co = row // CO, row % CO
                    t = e[3](rho)
                    if t is None:
                        continue
                    lane = kk * 16 + row
                    for ci in range(CI):
                        a[ci, s, lane] = widx(ci, t[0], t[1], t[2], co)
        aidx.append(a.reshape(-1))
        r0 += ks[q]
    aidx = np.concatenate(aidx)
    tpc_max = (8 if max(ks) <= 12 else 3) if nq == 1 else 4        # registers: one operand offset per (tile, k-step)
    hlo, hhi = int(dlt[:, :, 1].min()), int(dlt[:, :, 1].max())
    # Tile choice: a block = W waves on PD position planes x PHB position rows (x all PW columns).  Measured (tools/diag/mm_sweep.py,
    # MI355X): what pays is SEVERAL co-resident blocks per CU -- their input waits, prologues, barriers and epilogues interleave with
    # each other's matrix phases -- i.e. a small LDS footprint (single-buffered input, row slabs instead of whole planes) and at most
    # 3 tiles per wave (the 128-register instances: 4 waves per SIMD); then full accumulator columns, a small halo and large channel
    # chunks.  _MM_TUNED pins the bench geometries where the sweep found a better tile than this score.
    tuned = _MM_TUNED.get((mode, S, CI, CO, tuple(K), tuple(isz), tuple(osz))) if force is None else None
    pin = force or tuned
    cands = []
    for W in (8, 4):
        for PD in range(min(PDT, 16), 0, -1):
            LD = ld_of(PD)
            for nslab in (1, 2, 3, 4, 6, 8, 12):
                PHB = (PH + nslab - 1) // nslab
                if nslab > 1 and ((PH + PHB - 1) // PHB != nslab or PHB < 2):
                    continue
                npos = PD * PHB * PW
                tpc = (npos + W * 16 - 1) // (W * 16)
                if tpc > tpc_max:
                    continue
                rows_max = min((PHB - 1) * shi + (hhi - hlo) + 1, IH)
                whole = (nslab == 1 and hlo <= 0 and (PH - 1) * shi + hhi + 1 >= IH)
                LPH = IHW if whole else ((rows_max * IW + 3) // 4) * 4 + 4
                CHP = 64 + ((LD * LPH + 63) // 64) * 64
                bps = ((PDT + PD - 1) // PD) * nslab
                for cc in ([CI] if nq > 1 else [c for c in range(CI, 0, -1) if CI % c == 0]):
                    for dbuf in (1, 0):
                        if pin and (W, PD, PHB, cc, dbuf) != tuple(pin):
                            continue
                        lds = (((aidx.size + 63) // 64) * 64 + rows * 64 + (1 + dbuf) * cc * CHP + 64) * 4
                        if lds > _MM_LDS_BUDGET:
                            continue
                        per_simd = 4 if (tpc <= 3 or nq > 1) else 2                               # waves per SIMD the instance's registers allow
                        nblk = min(_MM_LDS_CU // lds, per_simd * 4 // W)
                        if nblk < 1:
                            continue
                        util = npos / float(tpc * W * 16) * (PDT / float(((PDT + PD - 1) // PD) * PD)) * (PH / float(nslab * PHB))   # filled accumulator columns
                        halo = ((PD * sdi) / float(LD)) * min(1.0, (PHB * shi) / float(rows_max))        # useful share of the staged elements
                        wps = nblk * W // 4                                                        # co-resident waves per SIMD
                        occ = {1: 0.45, 2: 0.6, 3: 0.8}.get(wps, 1.0)
                        # whole planes in 8-wave blocks measured best wherever they fit (41x49x35: row slabs / 4-wave blocks 3-20 % slower:
                        # more halo rows, fewer waves to balance a block's tiles over); slabs are what lets the 82x98x70 planes in at all
                        shape = (1.0 if nslab == 1 else 0.85) * (1.0 if W == 8 else 0.9)
                        score = util * (0.5 + 0.5 * halo) * (0.9 + 0.1 * cc / float(CI)) * occ * shape * (1.0 if (dbuf or nblk >= 2) else 0.9)
                        cands.append((score, PD, LD, cc, tpc, dbuf, PHB, W))
    if not cands:
        return None
    _, PD, LD, cc, tpc, dbuf, PHB, W = max(cands, key=lambda c: (round(c[0], 9), c[3], -c[5]))
    tpc = {1: (3 if tpc <= 3 else tpc if tpc <= 6 else 8), 4: 4}[nq]
    return MmPlan(CI=CI, CO=CO, ID=ID, IH=IH, IW=IW, OD=OD, OH=OH, OW=OW, nq=nq, ks=ks, PDT=PDT, PH=PH, PW=PW, PD=PD, sdi=sdi, shi=shi,
                  swi=swi, d0=d0, LD=LD, cc=cc, sdo=sdo, sho=sho, swo=swo, od0=od0, oh0=oh0, ow0=ow0, tpc=tpc, slack=slack, dbuf=dbuf,
                  PHB=PHB, hlo=hlo, hhi=hhi, waves=W,
                  tau=tau.reshape(-1), dlt=dlt.reshape(-1), aidx=aidx, mode=mode)


def conv_mm(x, plan: MmPlan, aimg, bias, relu_in, scale, shift, per_group, mask_src=None, next_bn=None):
    """One vg_conv_mm launch: x [N][CI][...] -> y [N][CO][OD][OH][OW] (pre-activation); with next_bn also the statistics partials."""
    lib = _lib.get_lib()
    N = x.shape[0]
    assert tuple(x.shape[1:]) == (plan.CI, plan.ID, plan.IH, plan.IW), (tuple(x.shape), plan.CI, plan.ID, plan.IH, plan.IW)
    tau, dlt, _ = plan.tables(x.device)
    d = plan.desc(N, relu_in, per_group if scale is not None else 1)
    y = torch.empty((N, plan.CO, plan.OD, plan.OH, plan.OW), dtype=torch.float32, device=x.device)
    _chk(x); _chk(aimg)
    if next_bn:
        chunks = lib.size('vg_conv_mm_stats_chunks', ctypes.byref(d), int(next_bn))
        G = N // int(next_bn)
        part = torch.zeros(G * plan.CO * chunks * 2, dtype=torch.float64, device=x.device)
        _call(x, 'vg_conv_mm', ctypes.byref(d), _p(x), _p(aimg), _p(tau), _p(dlt), _p(bias), _p(scale), _p(shift), _p(mask_src), _p(y),
              int(next_bn), 1, _p(part))
        return y, part
    _call(x, 'vg_conv_mm', ctypes.byref(d), _p(x), _p(aimg), _p(tau), _p(dlt), _p(bias), _p(scale), _p(shift), _p(mask_src), _p(y), 1, 0, None)
    return y
