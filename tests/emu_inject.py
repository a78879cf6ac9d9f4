"""TEST SEAM (lives under tests/, not in the product): swap the host build of the kernel sources (tests/emu,
g++ -DVG_EMU) in for libvaegam_hip.so by monkeypatching vae_gam_amd._lib._LIB.  The injected handle declares that it
accepts host pointers; the product library never does, so the product keeps refusing CPU tensors."""
import os
import subprocess

import vae_gam_amd  # noqa: F401
from vae_gam_amd import _lib

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu')


class EmuLibrary(_lib.VgLibrary):
    host_pointers_ok = True


def build_emu():
    so = os.path.join(EMU_DIR, 'libvaegam_emu.so')
    csrc = os.path.join(os.path.dirname(EMU_DIR), '..', 'vae-gam_amd', 'csrc')
    srcs = [os.path.join(EMU_DIR, f) for f in os.listdir(EMU_DIR) if f.endswith(('.h', '.cpp', '.sh'))]
    srcs += [os.path.join(csrc, f) for f in os.listdir(csrc)]
    srcs.append(os.path.join(os.path.dirname(EMU_DIR), '..', 'include', 'vaegam.h'))
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call([os.path.join(EMU_DIR, 'build_emu.sh')])
    return so


def inject_emu():
    """-> the previous handle (pass it to restore())."""
    prev = _lib._LIB
    _lib._LIB = EmuLibrary(build_emu())
    return prev


def use_product_library():
    """GPU tests: make sure no injected handle is left over (the next get_lib() loads libvaegam_hip.so)."""
    if _lib._LIB is not None and _lib._LIB.host_pointers_ok:
        _lib._LIB = None


def restore(prev):
    _lib._LIB = prev
