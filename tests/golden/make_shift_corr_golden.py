"""Mints tests/golden/shift_corr_ref.npz from the REFERENCE'S OWN correlation kernel.

The kernel (CorrelateData, /root/reference/Nets/Native/shift_corr.cu.cc:17-70) and its launcher (:193-233) are compiled from
the source where it lies by oracle/Makefile -- here the CPU build (_ref/libshift_corr_ref_cpu.so: the same source against the
functional emulator of tests/emul), so the fixture can be produced in a GPU-less container; tests/test_ref_pin.py re-runs the
GPU build (_ref/libshift_corr_ref.so) on the MI355X and checks that it reproduces these values.

Inputs are regenerated from the seed (numpy PCG64 `default_rng`, stable across numpy versions); the file stores the seed, a
CRC of the inputs and the kernel's NCHW output.  Layout contract: sharedLayers.correlation_native (Nets/sharedLayers.py:31-39)
-- in0 / in1 NHWC zero-padded by max_disp along W, out NCHW.

usage (needs /root/reference): make -C oracle && python tests/golden/make_shift_corr_golden.py
"""
import ctypes as C
import os
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

# (name, B, H, W, C, max_disp): the five MADNet cost-volume shapes (md = 2; the two largest levels cropped in H to keep the
# fixture small -- rows are independent) and the DispNet search range md = 40 at C = 128 on a 4-row, 128-pixel strip (both borders + interior), plus odd sizes.
CASES = [
    ("madnet_l6", 1, 6, 20, 192, 2),
    ("madnet_l5", 1, 12, 40, 128, 2),
    ("madnet_l4", 1, 24, 80, 96, 2),
    ("madnet_l3", 1, 24, 160, 64, 2),
    ("madnet_l2", 2, 12, 320, 32, 2),
    ("dispnet_md40", 1, 4, 128, 128, 40),
    ("odd", 2, 5, 17, 33, 3),
]


def make_inputs(name, B, H, W, Cc, md):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    L = rng.standard_normal((B, H, W, Cc)).astype(np.float32)
    R = rng.standard_normal((B, H, W, Cc)).astype(np.float32)
    return L, R


def run_reference(lib, L, R, md):
    """-> NCHW [B, 2*md+1, H, W] exactly as ShiftCorrKernelLauncher wrote it."""
    B, H, W, Cc = L.shape
    Lp = np.ascontiguousarray(np.pad(L, ((0, 0), (0, 0), (md, md), (0, 0))))
    Rp = np.ascontiguousarray(np.pad(R, ((0, 0), (0, 0), (md, md), (0, 0))))
    out = np.zeros((B, 2 * md + 1, H, W), np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.ref_shift_corr(p(Lp), p(Rp), md, B, H, W + 2 * md, Cc, p(out))
    assert rc == 0
    return out


def main():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libshift_corr_ref_cpu.so"))
    data = {}
    for name, B, H, W, Cc, md in CASES:
        L, R = make_inputs(name, B, H, W, Cc, md)
        data[name + "/out"] = run_reference(lib, L, R, md)
        data[name + "/crc"] = np.array([zlib.crc32(L.tobytes()), zlib.crc32(R.tobytes())], np.uint32)
    np.savez_compressed(os.path.join(HERE, "shift_corr_ref.npz"), **data)
    print("wrote shift_corr_ref.npz:", {k: v.shape for k, v in data.items() if k.endswith("/out")})


if __name__ == "__main__":
    main()
