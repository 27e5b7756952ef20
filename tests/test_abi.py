"""The C-ABI library loads and exports EVERY symbol include/madnet_hip.h (+ the tuning hooks of include/madnet_hip_tune.h) declares (no compute
calls: this runs without a GPU), and the product loader fails loudly without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "madnet_hip.h")
TUNE_HEADER = os.path.join(ROOT, "include", "madnet_hip_tune.h")


def _declared(paths=(HEADER, TUNE_HEADER)):
    names = set()
    for path in paths:
        src = open(path).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(mh_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_tuning_hooks_live_in_their_own_header():
    """nothing in the drop-in header is a benchmark knob (the hooks are not part of the reference interface)"""
    assert not [n for n in _declared((HEADER,)) if n.startswith("mh_tune_")]
    assert all(n.startswith("mh_tune_") for n in _declared((TUNE_HEADER,)))


def test_header_symbols_exported():
    import __graft_entry__  # noqa: F401  (sets sys.path)
    from madnet_hip import _ffi
    if not os.path.exists(_ffi.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    dll = ctypes.CDLL(_ffi.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(dll, n), "symbol %s declared in include/madnet_hip.h but not exported" % n
    assert set(names) == set(_ffi.SIGNATURES.keys()), set(names) ^ set(_ffi.SIGNATURES.keys())
    assert dll.mh_abi_version() == 16


def test_product_loader_fails_loudly_without_gpu():
    import torch
    from madnet_hip import _ffi
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    _ffi._lib = None
    with pytest.raises(_ffi.MadnetHipError):
        _ffi.lib()


def test_struct_layout_matches_header():
    from madnet_hip import _ffi
    assert ctypes.sizeof(_ffi.ConvDesc) == 24 * 4
    assert ctypes.sizeof(_ffi.Op) == 4 + 27 * 4 + 4 * 4 + 12 * 8 + 8
    assert ctypes.sizeof(_ffi.ShadowSeg) == 2 * 8 + 8 + 4 * 4 and ctypes.sizeof(_ffi.WgsLayer) == 4 * 8 + 14 * 4
    assert ctypes.sizeof(_ffi.PlaneSeg) == 3 * 8 + 8 + 4 * 4 + 8 + 2 * 4
