"""The C ABI: every entry point include/vaegam.h declares is exported by libvaegam_hip.so and bound by the
ctypes layer (no compute calls: this runs without a GPU)."""
import os
import re

import vae_gam_amd  # noqa: F401
from vae_gam_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'vaegam.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(vg_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_bound_and_exported():
    decl = declared_symbols()
    assert set(decl) == set(_lib.EXPORTS), (sorted(set(decl) ^ set(_lib.EXPORTS)))
    so = build.build_hip(verbose=False)
    lib = _lib.VgLibrary(so)                       # resolves every symbol or raises AttributeError
    assert lib.dll.vg_version() >= 100
    assert lib.dll.vg_last_error() is not None


def test_missing_library_fails_loudly(tmp_path):
    import pytest
    with pytest.raises(_lib.VgError, match='no CPU fallback'):
        _lib.VgLibrary(str(tmp_path / 'libvaegam_hip.so'))
