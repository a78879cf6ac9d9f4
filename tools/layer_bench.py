"""Per-layer kernel timings at the bench shapes (B=32, C=3): forward, data-gradient, weight-gradient.
Run on the GPU box:  python tools/layer_bench.py [B] [C]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vae_gam_amd
from vae_gam_amd import ops, _lib
if os.environ.get('VG_LIB'):
    _lib._LIB = _lib.VgLibrary(os.environ['VG_LIB'])     # diagnostic builds (ablations)
from vae_gam_amd.schema import net_geometry

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
C = int(sys.argv[2]) if len(sys.argv) > 2 else 3
geom = net_geometry((41, 49, 35))
dev = 'cuda'


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3     # us


tot = {'fwd': 0, 'bwd': 0, 'wgrad': 0}
print('%-8s %10s %10s %10s   %8s  (us; GFLOP per launch; TF/s of the slowest)' % ('layer', 'fwd', 'bwd-data', 'wgrad', 'GFLOP'))
for specs, sizes, N in ((geom.enc, geom.enc_sizes(), B), (geom.dec, geom.dec_sizes(), (C + 1) * B)):
    for i, sp in enumerate(specs):
        if os.environ.get('VG_LAYERS') and sp.name not in os.environ['VG_LAYERS'].split(','):
            continue
        x = torch.randn((N, sp.ci) + sizes[i], device=dev)
        wshape = ((sp.co, sp.ci) if sp.kind == 'conv' else (sp.ci, sp.co)) + tuple(sp.k)
        w = torch.randn(wshape, device=dev) * 0.1
        b = torch.zeros(sp.co, device=dev)
        sc = torch.ones((N // B) * sp.ci, device=dev); sh = torch.zeros((N // B) * sp.ci, device=dev)
        wf = ops.pack_weight(w, sp, 'fwd'); wb = ops.pack_weight(w, sp, 'bwd')
        y = ops.conv_forward(x, wf, b, sp, True, sc, sh, B)
        dy = torch.randn_like(y)
        mmf = ops._mm_for(None, w, sp, 'fwd', sizes[i], None)           # matrix-core kernel where a plan exists (VG_CONV_MM=0: off)
        mmb = ops._mm_for(None, w, sp, 'bwd', sizes[i + 1], sizes[i]) if sp.name != 'conv1' else None
        if mmf is not None:
            ym = ops.conv_mm(x, mmf[0], mmf[1], b, True, sc, sh, B)
            assert os.environ.get('VG_DBG') or float((ym - y).abs().max()) <= 1e-3 * float(y.abs().max()), sp.name
            t_f = timeit(lambda: ops.conv_mm(x, mmf[0], mmf[1], b, True, sc, sh, B))
        else:
            t_f = timeit(lambda: ops.conv_forward(x, wf, b, sp, True, sc, sh, B))
        if sp.name == 'conv1':
            t_b = 0.0
        elif mmb is not None:
            t_b = timeit(lambda: ops.conv_mm(dy, mmb[0], mmb[1], None, False, None, None, 1, x))
        else:
            t_b = timeit(lambda: ops.conv_backward_data(dy, wb, sp, sizes[i], x))
        tag = ('M' if mmf is not None else '-') + ('M' if mmb is not None else '-')
        t_w = timeit(lambda: ops.conv_weight_grad(x, dy, sp, True, sc, sh, B))
        macs = N * sp.co * sp.ci * int(np.prod(sp.k)) * int(np.prod(sizes[i + 1] if sp.kind == 'conv' else sizes[i]))
        gf = 2 * macs / 1e9
        tot['fwd'] += t_f; tot['bwd'] += t_b; tot['wgrad'] += t_w
        print('%-8s %10.1f %10.1f %10.1f   %8.2f  %6.2f TF/s  %s' % (sp.name, t_f, t_b, t_w, gf, gf / max(t_f, t_b, t_w) * 1e-3 * 1e3 / 1e3 * 1e3 / 1e3 if False else gf / (max(t_f, t_b, t_w) * 1e-6) / 1e3, tag))
print('total us: fwd %.0f  bwd-data %.0f  wgrad %.0f  sum %.0f' % (tot['fwd'], tot['bwd'], tot['wgrad'], sum(tot.values())))
