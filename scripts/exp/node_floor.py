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

