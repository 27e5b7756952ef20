"""Where the 0.1 ms between the replayed step (bench `value`) and the reference-FPS loop (`step_surface`) goes: Adapter.step with parts of its host side removed.
usage: python scripts/exp/step_surface_phases.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import Nets
from madnet_hip import _ffi, engine as E, synthetic as S
from madnet_hip.adapter import Adapter
from Data_utils.data_reader import device_prefetcher

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
H, W = 375, 1242
lib = _ffi.lib()
wn = S.calibrated_weights(dict(E.madnet_manifest()), 1)
pairs = [S.make_pair(H, W, stream_id=100, frame=t) for t in range(8)]
pairs8 = [(l.astype(np.uint8), r.astype(np.uint8), np.ascontiguousarray(g[..., 0])) for l, r, g in pairs]
z = torch.zeros(1, H, W, 3, device="cuda")
net = Nets.get_stereo_net("MADNet", {"left_img": z, "right_img": z, "split_layers": [None], "sequence": True, "train_portion": "BEGIN", "bulkhead": False, "weights": wn,
                                     "precision": "mixed", "warping": True, "context_net": True, "radius_d": 2, "stride": 1})
ad = Adapter(net, mode="FULL", lr=1e-4)
ad._plan("FULL")


class Source(object):
    def __init__(self, n): self.n = n
    def __iter__(self):
        for t in range(self.n):
            yield pairs8[t % 8]


def loop(name, frames_iter, n, warm=20):
    k, t0 = 0, None
    for f in frames_iter:
        if k == warm:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        ad.step(*f)
        k += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (k - warm)
    print("%-64s %8.1f us/step" % (name, dt * 1e6)); sys.stdout.flush()
    return dt


dev_frames = [tuple(torch.as_tensor(a, dtype=torch.float32, device="cuda").reshape(s) for a, s in zip(p, ((1, H, W, 3), (1, H, W, 3), (1, H, W)))) for p in pairs8]
def resident(n):
    for t in range(n):
        yield dev_frames[t % 8]

loop("prefetcher (the bench's step_surface loop)", device_prefetcher(Source(steps + 20), device="cuda", depth=3, consumer_stream=ad.stream), steps)
loop("frames resident in HBM (no reader thread, no H2D)", resident(steps + 20), steps)
up = ad._upload
ad._upload = lambda *a, **k: None
loop("  + no D2D upload into the engine's input buffers", resident(steps + 20), steps)
rb = ad._readback
ad._readback = lambda: None
loop("  + no read-back copies (sync only)", resident(steps + 20), steps)
fin = ad._finish
ad._finish = lambda: {}
loop("  + no host bookkeeping (_finish)", resident(steps + 20), steps)
ad._upload, ad._readback, ad._finish = up, rb, fin
# the replay alone, back to back without a sync per step
p = ad._plan("FULL")[0]
with torch.cuda.stream(ad.stream):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        p.launch(lib, ad.stream.cuda_stream)
    torch.cuda.synchronize()
print("%-64s %8.1f us/step" % ("graph replays back to back, one sync at the end", (time.perf_counter() - t0) / steps * 1e6))
with torch.cuda.stream(ad.stream):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        p.launch(lib, ad.stream.cuda_stream)
        ad.stream.synchronize()
    torch.cuda.synchronize()
print("%-64s %8.1f us/step" % ("graph replay + stream.synchronize() every step", (time.perf_counter() - t0) / steps * 1e6))
loop("prefetcher again", device_prefetcher(Source(steps + 20), device="cuda", depth=3, consumer_stream=ad.stream), steps)
