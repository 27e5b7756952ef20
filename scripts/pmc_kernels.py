"""Launches the two roofline kernels a few times (for rocprofv3 --pmc passes):
   conv 3x3 128->128 @ 96x320 (dil 2) and correlation fwd level-2 shape x 64 streams."""
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
L = torch.randn(64, 96, 320, 32, device=dev); R = torch.randn(64, 96, 320, 32, device=dev)
out = torch.empty(64, 96, 320, 5, device=dev)
for _ in range(5):
    ops.PRECISION = 0
    ops.conv2d_fwd(lib, ops.view(x), w, b, ops.view(y), dil=2, alpha=0.2, stream=0)
    ops.PRECISION = 1           # bf16 MFMA variant of the same layer
    ops.conv2d_fwd(lib, ops.view(x), w, b, ops.view(y), dil=2, alpha=0.2, stream=0)
    ops.PRECISION = 0
    ops.corr_fwd(lib, ops.view(L), ops.view(R), ops.view(out), 2, stream=0)
torch.cuda.synchronize()
print("done")
