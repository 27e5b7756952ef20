"""What does one dependent kernel node cost inside a replayed hipGraph on this runtime?  Chains of N identical trivial ops (a 1-float fill, a device
time stamp, a 4 MB channel copy, the small-layer conv of a 6x20 level) captured as one graph; us per node = (replay time of 2N nodes - replay time of N) / N.
The step has ~140 nodes: every us of per-node floor is 9 % of it.   usage: python scripts/exp/node_floor.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")):
    sys.path.insert(0, p)
import torch
from madnet_hip import _ffi, ops
from madnet_hip.plan import Recorder
from madnet_hip.benchtools import _time_ms
lib = _ffi.lib()
st = torch.cuda.Stream()
dev = "cuda"


def chain(rec_fn, n):
    r = Recorder()
    for i in range(n):
        rec_fn(r, i)
    return r.compile()


def per_node(name, rec_fn, n=64):
    res = []
    for m in (n, 2 * n):
        pl = chain(rec_fn, m)
        with torch.cuda.stream(st):
            pl.run(lib, st.cuda_stream); st.synchronize()
            pl.capture(lib, st.cuda_stream)
            for _ in range(3):
                pl.launch(lib, st.cuda_stream)
            st.synchronize()
            res.append(_time_ms(lib, st, lambda: pl.launch(lib, st.cuda_stream), 20) * 1e3)
    print("%-46s %3d nodes %8.1f us   %3d nodes %8.1f us   -> %.2f us per node (graph fixed part %.1f us)" % (name, n, res[0], 2 * n, res[1], (res[1] - res[0]) / n, 2 * res[0] - res[1]))


buf = torch.zeros(1 << 22, device=dev)
slots = torch.zeros(256, dtype=torch.int64, device=dev)
per_node("fill 1 float (1 workgroup)", lambda r, i: r.fill(C.c_void_p(buf.data_ptr()), 1, 0.0, None))
per_node("fill 4 MB (1024 workgroups)", lambda r, i: r.fill(C.c_void_p(buf.data_ptr()), 1 << 20, 0.0, None))
per_node("time stamp (1 lane)", lambda r, i: ops.stamp(r, slots, i % 256))
a = torch.randn(1, 96, 320, 32, device=dev); b = torch.zeros(1, 96, 320, 32, device=dev)
per_node("copy_channels 96x320x32 (3.9 MB)", lambda r, i: ops.copy_channels(r, ops.view(a if i % 2 == 0 else b), ops.view(b if i % 2 == 0 else a)))
# the small-layer bank kernel on a 6x20 level: 128 -> 128, forward bf16, ping-pong between two buffers (a dependent chain like the estimator's)
x0 = torch.randn(1, 6, 20, 128, device=dev) * 0.1; x1 = torch.zeros(1, 6, 20, 128, device=dev)
w = torch.randn(3, 3, 128, 128, device=dev) * 0.02; bias = torch.zeros(128, device=dev)
keep = []
bank = torch.zeros(ops.pack_bytes(w, 1, 0) // 4, device=dev)
ops.pack_weights(lib, [(w, bank, 1, 0)], dev, keep)
ops.PRECISION = 1
per_node("conv_bank_small 6x20 128->128 (bf16, 24 WGs)", lambda r, i: ops.conv2d_fwd(r, ops.view(x0 if i % 2 == 0 else x1), w, bias, ops.view(x1 if i % 2 == 0 else x0), alpha=0.2, wb=bank))
x0 = torch.randn(1, 24, 80, 128, device=dev) * 0.1; x1 = torch.zeros(1, 24, 80, 128, device=dev)
per_node("conv_bank_small 24x80 128->128 (bf16, 240 WGs)", lambda r, i: ops.conv2d_fwd(r, ops.view(x0 if i % 2 == 0 else x1), w, bias, ops.view(x1 if i % 2 == 0 else x0), alpha=0.2, wb=bank))
ops.PRECISION = 0


# ---- the plane kernel inside a replayed graph, phases removed (mh_tune_conv_planes bits 8 = no K walk, 9 = no staging, 12 = no epilogue, 13 = no stores):
# what a launch costs in the step, without the eager-launch floor of scripts/microbench.py phases
def planes_chain(label, B, H, W, Ci, Co, bits_list):
    x = torch.randn(B, H, W, Ci, device=dev) * 0.1
    w = torch.randn(3, 3, Ci, Co, device=dev) * 0.02; b = torch.zeros(Co, device=dev)
    keep = []
    bank32 = torch.zeros(ops.pack_bytes(w, 2, 2) // 4, device=dev)
    ops.pack_weights(lib, [(w, bank32, 2, 2)], dev, keep)
    xp = ops.Planes(ops.Shadow(B, H, W, Ci, dev), dev)
    yp = [ops.Planes(ops.Shadow(B, H, W, Co, dev), dev) for _ in range(2)]
    ops.plane_split(lib, [(ops.view(x), xp)], dev, keep)
    for bits, name in bits_list:
        lib.tune_conv_planes(bits << 8)
        try:
            # (the chain alternates two output plane sets; the input stays: the phases, not the data flow, are measured)
            per_node("planes %s: %s" % (label, name), lambda r, i: ops.conv2d_planes(r, xp, w, bank32, b, out=None, out_planes=yp[i % 2], alpha=0.2), n=32)
        finally:
            lib.tune_conv_planes(0)


PH = [(0, "full"), (1, "no K walk"), (16, "no epilogue"), (17, "no K walk, no epilogue"), (19, "no K walk, no staging, no epilogue"), (33, "no K walk, no stores")]
planes_chain("128->128 80x304", 1, 80, 304, 128, 128, PH)
planes_chain("38->128 80x304", 1, 80, 304, 38, 128, PH)
planes_chain("128->128 40x152", 1, 40, 152, 128, 128, PH)
