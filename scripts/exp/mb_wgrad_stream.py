"""Microbenchmark of the streaming filter-gradient kernel (csrc/wgrad_stream.hip) next to the tiled kernels (csrc/wgrad.hip) on the estimator-2 and
context-network batches of MADNet at 96 x 320 (B = 1 and 4): shadow cast / stream launch / split reduction timed separately with HIP events."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")):
    sys.path.insert(0, p)
import torch
from madnet_hip import _ffi, ops

lib = _ffi.lib()
dev = "cuda"


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


H, W = 96, 320
EST = [(38, 128, 1, 40), (128, 128, 1, 128), (128, 96, 1, 128), (96, 64, 1, 96), (64, 32, 1, 64), (32, 1, 1, 32)]
CTX = [(33, 128, 1, 36), (128, 128, 2, 128), (128, 128, 4, 128), (128, 96, 8, 128), (96, 64, 16, 96), (64, 32, 1, 64), (32, 1, 1, 32)]
ONE = [(128, 128, 1, 128)]
G3 = [(70, 128, 1, 72), (128, 128, 1, 128), (128, 96, 1, 128), (96, 64, 1, 96), (64, 32, 1, 64), (32, 1, 1, 32)]

for name, layers, hh, ww in (("one128", ONE, H, W), ("est2", EST, H, W), ("ctx", CTX, H, W), ("est3", G3, 48, 160)):
    for B in ((1,) if os.environ.get('MB_ONLY_B1') else (1, 4)):
        xs_f, zs_f, items, pairs, dws = [], [], [], [], []
        flops = 0.0
        for (Ci, Co, dil, ild) in layers:
            xb = torch.randn(B, hh, ww, ild, device=dev); gz = torch.randn(B, hh, ww, Co, device=dev)
            xv = ops.View(xb, B, hh, ww, Ci, ild)
            zv = ops.view(gz) if Co > 1 else ops.view(gz[..., 0].contiguous())
            xs, zs = ops.Shadow(B, hh, ww, Ci, dev), ops.Shadow(B, hh, ww, Co, dev)
            dw = torch.zeros(3, 3, Ci, Co, device=dev); db = torch.zeros(Co, device=dev)
            pairs += [(xv, xs), (zv, zs)]
            items.append((xs, zs, dw, db, dil)); dws.append((xv, zv, dw, db, dil)); xs_f.append(xb); zs_f.append(gz)
            flops += 2.0 * B * hh * ww * 9 * Ci * Co
        keep = []
        t_cast = timeit(lambda: ops.shadow_cast(lib, pairs, dev, keep))
        for nw, dist in ((8, 1), (6, 2), (4, 2), (4, 1)):
            if max(l[2] for l in layers) > 8 and nw > 7:
                nw = 7
            lib.tune_wgrad_stream(dist)
            for wgs in (256, 512) if nw <= 4 else (256,):
                wsa = ops.WgradWorkspace(dev); segs = []
                rec = []

                def run_stream():
                    wsa.reset(); del segs[:]
                    ops.wgrad_stream(lib, lib, wsa, segs, items, dev, keep, target_wgs=wgs, nwaves=nw)
                t_s = timeit(run_stream)
                kern = lib.last_kernel().decode()
                seg_snapshot = list(segs)
                t_r = timeit(lambda: ops.wgrad_reduce(lib, seg_snapshot, dev, keep)) if seg_snapshot else (0.0, 0.0)
                wsb = sum(sz * sp * 4 for _, _, sz, sp in seg_snapshot)
                print("%-7s B=%d nw=%d dist=%d wgs=%d | cast %6.1f us | stream %6.1f (min %6.1f) us = %6.1f TFLOP/s | reduce %5.1f us | ws %6.2f MB | %s"
                      % (name, B, nw, dist, wgs, t_cast[0], t_s[0], t_s[1], flops / t_s[0] * 1e-6, t_r[0], wsb / 1e6, kern))
        lib.tune_wgrad_stream(0)
        # the tiled kernels on the same layers (partial sums + one reduction), bf16 mode
        ops.PRECISION = 1
        try:
            wsa = ops.WgradWorkspace(dev); segs = []

            def run_tiled():
                wsa.reset(); del segs[:]
                for xv, zv, dw, db, dil in dws:
                    ops.conv2d_wgrad_partial(lib, lib, wsa, segs, xv, zv, dw, db, dil=dil)
            t_t = timeit(run_tiled)
            snap = list(segs)
            t_tr = timeit(lambda: ops.wgrad_reduce(lib, snap, dev, keep))
            wsb = sum(sz * sp * 4 for _, _, sz, sp in snap)
            print("%-7s B=%d tiled kernels (%d launches)            | partial %6.1f us = %6.1f TFLOP/s | reduce %5.1f us | ws %6.2f MB"
                  % (name, B, len(dws), t_t[0], flops / t_t[0] * 1e-6, t_tr[0], wsb / 1e6))
        finally:
            ops.PRECISION = 0
        del keep[:]
