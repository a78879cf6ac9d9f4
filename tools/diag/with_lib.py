"""Run the GPU tests or bench.py against a diagnostic build of the library (tools/diag/build_variant.sh / build_all_variant.sh):
  python tools/diag/with_lib.py tools/diag/libvg_<name>.so pytest tests/test_kernels_gpu.py -x -q
  python tools/diag/with_lib.py tools/diag/libvg_<name>.so bench.py --no-cpu-baseline --kernel-table"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vae_gam_amd                                               # noqa: E402
from vae_gam_amd import _lib                                     # noqa: E402
_lib._LIB = _lib.VgLibrary(os.path.join(ROOT, sys.argv[1]) if not os.path.isabs(sys.argv[1]) else sys.argv[1])
what, args = sys.argv[2], sys.argv[3:]
if what == 'pytest':
    import pytest
    sys.exit(pytest.main(args))
sys.argv = [what] + args
runpy.run_path(os.path.join(ROOT, what), run_name='__main__')
