"""Launches the round-2 roofline kernels a few times each (for the rocprofv3 --pmc passes of scripts/gpu_pmc_r02.sh), all on the
layer / shapes bench.py quotes: 3x3 128->128 @ 96x320 dil 2 (context-2) forward in split-bf16 (x3 patch kernel) and in bf16, its
input gradient (bf16 patch kernel), its filter gradient (wgrad_bf16 partial sums), the MADNet correlation at B=64 and the DispNet
81-shift correlation at B=16 in bf16 MFMA."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")):
    sys.path.insert(0, p)
import torch
from madnet_hip import _ffi, ops

lib = _ffi.lib()
dev = "cuda"
x = torch.randn(1, 96, 320, 128, device=dev); w = torch.randn(3, 3, 128, 128, device=dev) * 0.05
b = torch.randn(128, device=dev); y = torch.empty(1, 96, 320, 128, device=dev)
dz = torch.randn(1, 96, 320, 128, device=dev); dx = torch.empty(1, 96, 320, 128, device=dev)
dw = torch.empty_like(w); db = torch.zeros(128, device=dev)
wsa = ops.WgradWorkspace(dev)
L = torch.randn(64, 96, 320, 32, device=dev); R = torch.randn(64, 96, 320, 32, device=dev); out = torch.empty(64, 96, 320, 5, device=dev)
L2 = torch.randn(16, 96, 320, 128, device=dev); R2 = torch.randn(16, 96, 320, 128, device=dev); out2 = torch.empty(16, 96, 320, 81, device=dev)
bank = torch.zeros(ops.pack_bytes(w) // 4, device=dev); keep = []
ops.pack_weights(lib, [(w, bank)], dev, keep)
xs = torch.randn(1, 24, 80, 128, device=dev); ys = torch.empty(1, 24, 80, 128, device=dev)          # 1/16 resolution: the small-layer bank kernel
bank1 = torch.zeros(ops.pack_bytes(w, 1) // 4, device=dev)
ops.pack_weights(lib, [(w, bank1, 1, 0)], dev, keep)
for _ in range(5):
    ops.conv2d_fwd(lib, ops.view(x), w, b, ops.view(y), dil=2, alpha=0.2, stream=0, precision=2, wb=bank)
    ops.conv2d_fwd(lib, ops.view(xs), w, b, ops.view(ys), alpha=0.2, stream=0, precision=1, wb=bank1)
    ops.conv2d_fwd(lib, ops.view(x), w, b, ops.view(y), dil=2, alpha=0.2, stream=0, precision=2)
    ops.conv2d_fwd(lib, ops.view(x), w, b, ops.view(y), dil=2, alpha=0.2, stream=0, precision=1)
    ops.PRECISION = 1
    ops.conv2d_dgrad(lib, ops.view(dz), w, ops.view(dx), dil=2, mask_ref=ops.view(x), mask_alpha=0.2, stream=0)
    wsa.reset(); segs = []
    ops.conv2d_wgrad_partial(lib, lib, wsa, segs, ops.view(x), ops.view(dz), dw, db, dil=2, stream=0)
    ops.PRECISION = 0
    ops.corr_fwd(lib, ops.view(L), ops.view(R), ops.view(out), 2, stream=0)
    ops.corr_fwd(lib, ops.view(L2), ops.view(R2), ops.view(out2), 40, stream=0, precision=1)
torch.cuda.synchronize()
print("done")
