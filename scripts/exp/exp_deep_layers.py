"""How much would split-K over workgroups buy on DispNet's skinny-M / long-K layers?  Times the real layer and the same
layer with K / 4 and K / 8 input channels (what one of 4 / 8 K-splits would run).  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")):
    sys.path.insert(0, p)
import torch
from madnet_hip import _ffi, ops
from madnet_hip.benchtools import _time_ms
lib = _ffi.lib(); stream = torch.cuda.Stream(); dev = "cuda"
ops.PRECISION = 1
for name, H, W, Ci, Co in (("conv6/1 1024->1024 @6x20", 6, 20, 1024, 1024), ("conv5/1 512->512 @12x40", 12, 40, 512, 512),
                           ("conv4/1 512->512 @24x80", 24, 80, 512, 512), ("conv6 512->1024 s1 @6x20", 6, 20, 512, 1024)):
    row = []
    for div in (1, 2, 4, 8):
        ci = Ci // div
        x = torch.randn(1, H, W, ci, device=dev); w = torch.randn(3, 3, ci, Co, device=dev) * 0.02; b = torch.randn(Co, device=dev)
        y = torch.empty(1, H, W, Co, device=dev)
        with torch.cuda.stream(stream):
            t = _time_ms(lib, stream, lambda: ops.conv2d_fwd(lib, ops.view(x), w, b, ops.view(y), alpha=0.1, stream=stream.cuda_stream), 20) * 1e3
        row.append("K/%d %.1f us" % (div, t))
    print("%-28s %s" % (name, "   ".join(row)))
