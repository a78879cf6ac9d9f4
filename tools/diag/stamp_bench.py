"""Where conv_mm's waves spend their cycles (diagnostic build with in-kernel stamps, tools/diag/build_stamp.sh).
Run on the GPU box:  python tools/diag/stamp_bench.py [B] [C] [layer:dir ...]   e.g. convt4:fwd convt3:fwd convt3:bwd
Prints, per launch, the mean share of every phase over all waves.  Read SHARES, not the length (the stamps fence overlaps)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import vae_gam_amd
from vae_gam_amd import ops, _lib
_lib._LIB = _lib.VgLibrary(os.path.join(ROOT, 'tools', 'diag', 'libvg_stamp.so'))
from vae_gam_amd.schema import net_geometry

args = sys.argv[1:]
B = int(args[0]) if args else 64
C = int(args[1]) if len(args) > 1 else 8
which = args[2:] or ['convt4:fwd', 'convt3:fwd', 'convt3:bwd', 'convt1:fwd', 'convt2:bwd']
geom = net_geometry((41, 49, 35))
dev = 'cuda'
SEG = ['dma_wait', 'prologue', 'barrier', 'stage_issue', 'matrix', 'store', 'stats_flush', 'loop_top']
rd = _lib._LIB.dll.vg_stamp_read
rd.restype = ctypes.c_int
rd.argtypes = [ctypes.c_void_p, ctypes.c_int]


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
for item in which:
    name, direction = item.split(':')
    sp, i, part = layers[name]
    sizes = geom.enc_sizes() if part == 'enc' else geom.dec_sizes()
    N = B if part == 'enc' else (C + 1) * B
    x = torch.randn((N, sp.ci) + sizes[i], device=dev)
    wshape = ((sp.co, sp.ci) if sp.kind == 'conv' else (sp.ci, sp.co)) + tuple(sp.k)
    w = torch.randn(wshape, device=dev) * 0.1
    b = torch.zeros(sp.co, device=dev)
    sc = torch.ones((N // B) * sp.ci, device=dev); sh = torch.zeros((N // B) * sp.ci, device=dev)
    force = tuple(int(v) for v in os.environ['VG_FORCE'].split(',')) if os.environ.get('VG_FORCE') else None
    if force:
        pl_ = ops.mm_plan(sp, direction, sizes[i] if direction == 'fwd' else sizes[i + 1], None if direction == 'fwd' else sizes[i], force=force)
    if direction == 'fwd':
        mm = ops._mm_for(None, w, sp, 'fwd', sizes[i], None) if not force else (pl_, pl_.gather(w))
        fn = lambda: ops.conv_mm(x, mm[0], mm[1], b, True, sc, sh, B, None, (B if name in ('convt2', 'convt4') and not os.environ.get('VG_NOSTATS') else None))
    else:
        dy = torch.randn((N, sp.co) + sizes[i + 1], device=dev)
        mm = ops._mm_for(None, w, sp, 'bwd', sizes[i + 1], sizes[i]) if not force else (pl_, pl_.gather(w))
        fn = lambda: ops.conv_mm(dy, mm[0], mm[1], None, False, None, None, 1, x)
    if mm is None:
        print(item, 'no plan'); continue
    pl = mm[0]
    t = timeit(fn)
    fn(); torch.cuda.synchronize()
    bps = (pl.PDT + pl.PD - 1) // pl.PD
    ns = max(1, min(256 // bps, N))
    grid = bps * ns
    buf = np.zeros(1024 * 16 * 8, np.uint64)
    rc = rd(buf.ctypes.data, buf.size)
    assert rc == 0, rc
    a = buf.reshape(1024, 16, 8)
    grid = int(np.nonzero(a.sum((1, 2)))[0].max()) + 1                 # blocks that ran (the library sizes the grid by occupancy)
    a = a[:grid, :8].astype(np.float64)
    tot = a.sum(-1)
    print('%s  %.1f us (stamped build)  grid %d  PD %d LD %d cc %d tpc %d ks %s  cycles/wave %.0f (min %.0f max %.0f)' % (
        item, t, grid, pl.PD, pl.LD, pl.cc, pl.tpc, list(pl.ks), tot.mean(), tot.min(), tot.max()))
    sh_ = a.sum((0, 1)) / a.sum()
    print('   ' + '  '.join('%s %.1f%%' % (SEG[k], 100 * sh_[k]) for k in range(8)))
    # by wave index (tile imbalance)
    bw = a.sum(0)
    print('   matrix share by wave: ' + ' '.join('%.0f%%' % (100 * bw[w, 4] / bw[w].sum()) for w in range(8)))
    print('   barrier share by wave: ' + ' '.join('%.0f%%' % (100 * bw[w, 2] / bw[w].sum()) for w in range(8)))
