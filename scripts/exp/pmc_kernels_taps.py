"""Launches the all-taps filter-gradient kernel (and the tiled kernel beside it) a few times on the 3x3 128->128 @ 96x320 layer, B = 1 and B = 4, for the
rocprofv3 --pmc passes of scripts/gpu_pmc_taps.sh: which stall bounds the barrier-coupled walk (0.59 us per 32-pixel segment with loads, MFMAs and stores
switched off; profiles/r02_microbench_wgrad_taps.txt)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")):
    sys.path.insert(0, p)
import torch
from madnet_hip import _ffi, ops

lib = _ffi.lib()
dev = "cuda"
ops.PRECISION = 1
for B in (1, 4):
    x = torch.randn(B, 96, 320, 128, device=dev); dz = torch.randn(B, 96, 320, 128, device=dev)
    dw = torch.empty(3, 3, 128, 128, device=dev); db = torch.zeros(128, device=dev)
    wsa = ops.WgradWorkspace(dev)
    for taps in (0, 1, 1 + 16 * 13):              # tiled | taps | taps skeleton (no loads, no MFMA walk, no partial stores)
        lib.tune_wgrad_taps(taps)
        for _ in range(5):
            wsa.reset(); segs = []
            ops.conv2d_wgrad_partial(lib, lib, wsa, segs, ops.view(x), ops.view(dz), dw, db, stream=0)
        torch.cuda.synchronize()
lib.tune_wgrad_taps(-1)
ops.PRECISION = 0
print("done")
