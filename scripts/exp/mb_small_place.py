"""Workgroup placement of the small-layer bank kernel (mh_tune_conv_bank_small): us per node of a dependent chain of the layer inside a replayed hipGraph,
for the estimator shapes of levels 6 / 5 / 4 / 3-pyramid, forward (bf16) and input gradient, under the four placement modes.
usage: python scripts/exp/mb_small_place.py"""
import os, sys
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


def per_node(rec_fn, n=64):
    res = []
    for m in (n, 2 * n):
        r = Recorder()
        for i in range(m):
            rec_fn(r, i)
        pl = r.compile()
        with torch.cuda.stream(st):
            pl.run(lib, st.cuda_stream); st.synchronize()
            pl.capture(lib, st.cuda_stream)
            for _ in range(3):
                pl.launch(lib, st.cuda_stream)
            st.synchronize()
            res.append(_time_ms(lib, st, lambda: pl.launch(lib, st.cuda_stream), 20) * 1e3)
    return (res[1] - res[0]) / n


MODES = [(0, "r5: all XCDs, pixel-major"), (1, "order model, all XCDs"), (2, "confined, pixel-major"), (3, "model (default)")]
print("%-34s %s" % ("layer", "   ".join("%-26s" % m[1] for m in MODES)))
for (H, W, Ci, Co) in [(6, 20, 197, 128), (6, 20, 128, 128), (6, 20, 64, 32), (12, 40, 134, 128), (12, 40, 128, 128), (12, 40, 96, 64), (24, 80, 102, 128), (24, 80, 128, 128),
                       (24, 80, 128, 96), (24, 80, 64, 32), (48, 160, 64, 96), (12, 40, 192, 192), (6, 20, 192, 192)]:
    w = torch.randn(3, 3, Ci, Co, device=dev) * 0.02; bias = torch.zeros(Co, device=dev)
    keep = []
    bank = torch.zeros(ops.pack_bytes(w, 1, 0) // 4, device=dev); bank_t = torch.zeros(ops.pack_bytes(w, 1, 1) // 4, device=dev)
    ops.pack_weights(lib, [(w, bank, 1, 0), (w, bank_t, 1, 1)], dev, keep)
    xs = [torch.randn(1, H, W, (Ci + 3) // 4 * 4, device=dev) * 0.1 for _ in range(2)]
    ys = [torch.zeros(1, H, W, (Co + 3) // 4 * 4, device=dev) for _ in range(2)]
    xv = [ops.View(t, 1, H, W, Ci, t.shape[-1]) for t in xs]; yv = [ops.View(t, 1, H, W, Co, t.shape[-1]) for t in ys]
    for what in ("fwd", "dgrad"):
        row = []
        for mode, _ in MODES:
            lib.tune_conv_bank_small(mode)
            ops.PRECISION = 1; ops.PRECISION_BWD = 1
            try:
                if what == "fwd":
                    t = per_node(lambda r, i: ops.conv2d_fwd(r, xv[i % 2], w, bias, yv[i % 2], alpha=0.2, wb=bank))
                else:
                    t = per_node(lambda r, i: ops.conv2d_dgrad(r, yv[i % 2], w, xv[i % 2], mask_ref=xv[(i + 1) % 2], mask_alpha=0.2, wb=bank_t))
            finally:
                ops.PRECISION = 0; ops.PRECISION_BWD = None
                lib.tune_conv_bank_small(-1)
            row.append(t)
        print("%-34s %s" % ("%s %dx%d %d->%d" % (what, H, W, Ci, Co), "   ".join("%-26s" % ("%.2f us" % t) for t in row)))
