"""Build libvaegam_hip.so (gfx950) in-tree with hipcc.  `python -m vae_gam_amd.build` or
`__graft_entry__.build()`.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['vg_api.hip', 'vg_conv.hip', 'vg_conv_mfma.hip', 'vg_wgrad.hip', 'vg_bn.hip', 'vg_gam.hip', 'vg_chol.hip', 'vg_latent.hip']
OUT = os.path.join(HERE, 'libvaegam_hip.so')


def hipcc_path():
    for c in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found (expected /opt/rocm/bin/hipcc)')


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'vaegam.h')]
    return all(os.path.getmtime(d) <= t for d in deps)


def build_hip(force=False, verbose=True):
    if not force and up_to_date():
        return OUT
    cmd = [hipcc_path(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           '-Wno-unused-result'] + [os.path.join(CSRC, s) for s in SOURCES] + ['-o', OUT]
    if verbose:
        print('[vae_gam_amd.build]', ' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build_hip(force='--force' in sys.argv)
