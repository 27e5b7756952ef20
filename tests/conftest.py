"""pytest configuration: `gpu` marker, import paths, backends.

Backends for the op-level parity tests:
  * "hip"  -- the product library on a real MI355X (tests marked `gpu`), torch CUDA tensors;
  * "emul" -- tests/emul/libmadnet_emul.so: the SAME kernel sources compiled against a CPU
              functional emulator (test infrastructure; never loaded by the product), torch CPU
              tensors.  Lets index/mask/tile logic be checked without a GPU.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: long CPU test")


class Backend(object):
    def __init__(self, name, lib, device):
        self.name, self.lib, self.device = name, lib, device

    def sync(self):
        if self.device != "cpu":
            import torch
            torch.cuda.synchronize()


_cache = {}


def _emul_backend():
    if "emul" not in _cache:
        from madnet_hip import _ffi
        d = os.path.join(ROOT, "tests", "emul")
        r = subprocess.run(["make", "-C", d, "-j8"], capture_output=True, text=True)
        if r.returncode != 0:
            pytest.skip("emulator build failed: " + r.stderr[-400:])
        lib = _ffi.Lib(os.path.join(d, "libmadnet_emul.so"))
        lib.ensure_init()
        _cache["emul"] = Backend("emul", lib, "cpu")
    return _cache["emul"]


def _hip_backend():
    if "hip" not in _cache:
        import torch
        from madnet_hip import _ffi
        assert torch.cuda.is_available(), "gpu test selected but torch sees no GPU"
        _cache["hip"] = Backend("hip", _ffi.lib(), "cuda")
    return _cache["hip"]


BACKENDS = [pytest.param("emul", id="emul"), pytest.param("hip", marks=pytest.mark.gpu, id="hip")]


@pytest.fixture(params=BACKENDS)
def backend(request):
    return _emul_backend() if request.param == "emul" else _hip_backend()


@pytest.fixture
def hip():
    return _hip_backend()
