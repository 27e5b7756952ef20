"""Stand-alone timing of DispNet's iconv5 / iconv4 / iconv3 forward layers (bf16, concat rows of 1024+1 / 768+1 / 384+1 channels) on the
ragged uniform-tap instances against the generic loader (mh_tune_conv_tile bit 19), and of the same layers with K rounded down to a multiple of 64
(what the plain uniform-tap loader does on an aligned row)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "real-time-self-adaptive-deep-stereo_amd"))
import torch
from madnet_hip import _ffi, ops

lib = _ffi.lib()
dev = "cuda:0"


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


ops.PRECISION = 1
for (H, W, K, N) in ((12, 40, 1025, 512), (24, 80, 769, 256), (48, 160, 385, 128)):
    for (Kc, ld, tag) in ((K, (K + 3) // 4 * 4, "concat row"), (K - 1, K - 1, "aligned, K-1"), (K - 1, (K + 3) // 4 * 4, "K-1 in the concat row")):
        buf = torch.randn(1, H, W, ld, device=dev)
        xv = ops.View(buf, 1, H, W, Kc, ld)
        w = torch.randn(3, 3, Kc, N, device=dev) * 0.05
        b = torch.zeros(N, device=dev)
        y = torch.empty(1, H, W, N, device=dev)
        res = []
        for off in (0, 1):
            lib.tune_conv_tile(off << 19, 0)
            t = timed(lambda: ops.conv2d_fwd(lib, xv, w, b, ops.view(y), alpha=0.2))
            res.append((t, lib.last_kernel().decode()))
        lib.tune_conv_tile(0, 0)
        fl = 2.0 * H * W * 9 * Kc * N
        print("%dx%d K=%d N=%d %-22s default %.1f us (%.0f TF/s) %s | ragged off %.1f us %s" % (H, W, Kc, N, tag, res[0][0], fl / res[0][0] / 1e6, res[0][1], res[1][0], res[1][1]))
