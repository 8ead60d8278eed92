"""CPU-side checks of the C-ABI boundary: the built library loads and exports every symbol that
include/mvip_nerf.h declares, and the Python binding table matches the header.  No GPU calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'mvip_nerf.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(mvip_[a-z0-9_]+)\s*\(', txt)))


def test_header_matches_binding_table():
    from mvip_nerf_amd import _lib
    assert header_symbols() == sorted(_lib.DECLARED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    from mvip_nerf_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from mvip_nerf_amd.csrc.build import build
        build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert _lib.load().mvip_abi_version() == _lib.ABI_VERSION == 5
    assert _lib.load().mvip_build_is_experiment() == 0       # the product library is never a timing-experiment build
    assert _lib.load().mvip_mlp_packed_floats() == 597248
    assert _lib.load().mvip_strerror(-1).decode().startswith('invalid')


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of silently computing elsewhere."""
    import torch
    from mvip_nerf_amd import _lib
    with pytest.raises(_lib.MvipError):
        _lib.ptr(torch.zeros(4))
