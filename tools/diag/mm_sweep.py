"""Sweep conv_mm's tile choice (PD planes per block, cc channels per chunk) for one launch:  python tools/diag/mm_sweep.py B C layer:dir [lib]
Prints us per launch for every (PD, cc) the kernel accepts, with LDS bytes and blocks per CU the occupancy query reports."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import vae_gam_amd
from vae_gam_amd import ops, _lib
from vae_gam_amd.schema import net_geometry
B, C = int(sys.argv[1]), int(sys.argv[2])
geom = net_geometry((41, 49, 35))
dev = 'cuda'


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


layers = {sp.name: (sp, i, 'enc') for i, sp in enumerate(geom.enc)}
layers.update({sp.name: (sp, i, 'dec') for i, sp in enumerate(geom.dec)})
for item in sys.argv[3:]:
    name, direction = item.split(':')
    sp, i, part = layers[name]
    sizes = geom.enc_sizes() if part == 'enc' else geom.dec_sizes()
    N = B if part == 'enc' else (C + 1) * B
    x = torch.randn((N, sp.ci) + sizes[i], device=dev)
    wshape = ((sp.co, sp.ci) if sp.kind == 'conv' else (sp.ci, sp.co)) + tuple(sp.k)
    w = torch.randn(wshape, device=dev) * 0.1
    b = torch.zeros(sp.co, device=dev)
    sc = torch.ones((N // B) * sp.ci, device=dev); sh = torch.zeros((N // B) * sp.ci, device=dev)
    dy = torch.randn((N, sp.co) + sizes[i + 1], device=dev)
    base = ops.mm_plan(sp, direction, sizes[i] if direction == 'fwd' else sizes[i + 1], None if direction == 'fwd' else sizes[i])
    print('%s: default PD %d cc %d tpc %d' % (item, base.PD, base.cc, base.tpc))
    ci = base.CI
    aimg0 = base.gather(w)
    if direction == 'fwd':
        ref = ops.conv_mm(x, base, aimg0, b, True, sc, sh, B, None, None)
    else:
        ref = ops.conv_mm(dy, base, aimg0, None, False, None, None, 1, x)
    shi = base.shi
    for W in (8, 4):
      for PD in range(1, 9):
        for nslab in (1, 2, 3, 4, 6):
          PHB = (base.PH + nslab - 1) // nslab
          if nslab > 1 and (base.PH + PHB - 1) // PHB != nslab:
              continue
          for cc in [c for c in range(ci, 0, -1) if ci % c == 0][:3]:
            for dbuf in (0, 1):
                pl = ops.mm_plan(sp, direction, sizes[i] if direction == 'fwd' else sizes[i + 1], None if direction == 'fwd' else sizes[i], force=(W, PD, PHB, cc, dbuf))
                if pl is None:
                    continue
                aimg = pl.gather(w)
                if direction == 'fwd':
                    fn = lambda: ops.conv_mm(x, pl, aimg, b, True, sc, sh, B, None, None)
                else:
                    fn = lambda: ops.conv_mm(dy, pl, aimg, None, False, None, None, 1, x)
                try:
                    out = fn()
                    err = float((out - ref).abs().max())
                    t = timeit(fn, 5)
                except Exception as e:
                    print('  W %d PD %d PHB %d cc %d db %d: %s' % (W, PD, PHB, cc, dbuf, str(e)[:80])); continue
                print('  W %d PD %d PHB %2d cc %2d db %d tpc %d  %8.1f us   maxdiff %.2e' % (W, PD, PHB, cc, dbuf, pl.tpc, t, err), flush=True)
