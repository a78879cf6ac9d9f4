"""Where wgrad_rows_k's waves spend their cycles (diagnostic build: tools/diag/build_variant.sh wgstamp vg_wgrad.hip -DVG_STAMP).
  python tools/diag/stamp_wgrad.py [B] [C] [layers...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import vae_gam_amd
from vae_gam_amd import ops, _lib
_lib._LIB = _lib.VgLibrary(os.path.join(ROOT, 'tools', 'diag', 'libvg_wgstamp.so'))
from vae_gam_amd.schema import net_geometry
args = sys.argv[1:]
B = int(args[0]) if args else 64
C = int(args[1]) if len(args) > 1 else 8
which = args[2:] or ['convt4', 'convt3', 'convt5', 'convt2']
geom = net_geometry((41, 49, 35))
SEG = ['item_setup', 'barrier', 'copy_issue', 'copy_wait', 'copy_issue_next', 'matrix', 'copy_wait_ch', 'barrier_ch']
rd = _lib._LIB.dll.vg_stamp_read_wg
rd.restype = ctypes.c_int; rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
layers = {sp.name: (sp, i, 'enc') for i, sp in enumerate(geom.enc)}
layers.update({sp.name: (sp, i, 'dec') for i, sp in enumerate(geom.dec)})
for name in which:
    sp, i, part = layers[name]
    sizes = geom.enc_sizes() if part == 'enc' else geom.dec_sizes()
    N = B if part == 'enc' else (C + 1) * B
    x = torch.randn((N, sp.ci) + sizes[i], device='cuda')
    dy = torch.randn((N, sp.co) + sizes[i + 1], device='cuda')
    sc = torch.ones((N // B) * sp.ci, device='cuda'); sh = torch.zeros((N // B) * sp.ci, device='cuda')
    fn = lambda: ops.conv_weight_grad(x, dy, sp, True, sc, sh, B)
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e3
    buf = np.zeros(2048 * 4 * 8, np.uint64)
    assert rd(buf.ctypes.data, buf.size) == 0
    a = buf.reshape(2048, 4, 8)
    nz = np.nonzero(a.sum((1, 2)))[0]
    grid = int(nz.max()) + 1 if len(nz) else 0
    a = a[:grid].astype(np.float64)
    tot = a.sum(-1)
    print('%s wgrad %.1f us (stamped)  grid %d  cycles/wave %.0f (min %.0f max %.0f)' % (name, t, grid, tot.mean(), tot.min(), tot.max()))
    s_ = a.sum((0, 1)) / a.sum()
    print('   ' + '  '.join('%s %.1f%%' % (SEG[k], 100 * s_[k]) for k in range(8)))
