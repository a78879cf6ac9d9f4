"""Build libvaegam_hip.so (gfx950) in-tree with hipcc.  `python -m vae_gam_amd.build` or
`__graft_entry__.build()`.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['vg_api.hip', 'vg_conv.hip', 'vg_wgrad.hip', 'vg_bn.hip', 'vg_gam.hip', 'vg_chol.hip', 'vg_latent.hip', 'vg_gp.hip', 'vg_conv_mm.hip', 'vg_fc.hip']
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
    """One hipcc -c per source (in parallel: vg_wgrad.hip alone is ~3 minutes of template instances), then one link."""
    if not force and up_to_date():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    hipcc = hipcc_path()
    objdir = os.path.join(HERE, '_obj')
    os.makedirs(objdir, exist_ok=True)
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result']

    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + [os.path.join(HERE, '..', 'include', 'vaegam.h')]
    hdr_t = max(os.path.getmtime(h) for h in headers)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + '.o')
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(hdr_t, os.path.getmtime(os.path.join(CSRC, src))):
            return obj                                   # object newer than its source and every header: keep it
        cmd = [hipcc] + flags + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print('[vae_gam_amd.build]', ' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(4, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, '--offload-arch=gfx950', '-fPIC', '-shared'] + objs + ['-o', OUT]
    if verbose:
        print('[vae_gam_amd.build]', ' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build_hip(force='--force' in sys.argv)
