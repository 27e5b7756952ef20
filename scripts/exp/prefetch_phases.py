"""The prefetcher's share of the reference-FPS loop (step_surface_phases.py: +70 .. 100 us per step over frames resident in HBM): time blocked in the iterator vs in Adapter.step,
and variants of the hand-over.   usage: python scripts/exp/prefetch_phases.py [steps]"""
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
    def __init__(self, n, delay=0.0): self.n, self.delay = n, delay
    def __iter__(self):
        for t in range(self.n):
            if self.delay: time.sleep(self.delay)
            yield pairs8[t % 8]


def loop(name, it, warm=20):
    k, t0, tg, ts = 0, None, 0.0, 0.0
    it = iter(it)
    while True:
        a = time.perf_counter()
        try:
            f = next(it)
        except StopIteration:
            break
        b = time.perf_counter()
        if k == warm:
            torch.cuda.synchronize(); t0 = time.perf_counter(); tg = ts = 0.0; a = b = t0
        ad.step(*f)
        c = time.perf_counter()
        tg += b - a; ts += c - b
        k += 1
    torch.cuda.synchronize()
    n = k - warm
    print("%-70s %8.1f us/step   iterator %6.1f   Adapter.step %7.1f" % (name, (time.perf_counter() - t0) / n * 1e6, tg / n * 1e6, ts / n * 1e6)); sys.stdout.flush()


N = steps + 20
loop("prefetcher as shipped (cast=True)", device_prefetcher(Source(N), device="cuda", depth=3, consumer_stream=ad.stream))
loop("prefetcher as shipped, cast=False", device_prefetcher(Source(N), device="cuda", depth=3, consumer_stream=ad.stream, cast=False))
sw = sys.getswitchinterval()
sys.setswitchinterval(1e-4)
loop("  switch interval 0.1 ms (default %.0f ms)" % (sw * 1e3), device_prefetcher(Source(N), device="cuda", depth=3, consumer_stream=ad.stream))
sys.setswitchinterval(sw)
loop("  source paced at one frame per 1.2 ms (a camera, not a backlog)", device_prefetcher(Source(N, 0.0012), device="cuda", depth=3, consumer_stream=ad.stream))
loop("  depth 8", device_prefetcher(Source(N), device="cuda", depth=8, consumer_stream=ad.stream))


class NoCast(device_prefetcher):            # the reader uploads but never launches the cast kernels: is it the copy stream's kernels inside the step?
    def _reader(self):
        t = self._torch
        for arrays in self._ds:
            i = self._slot(arrays)
            host, devb, ev, stage, _ = self._ring[i]
            for h, a in zip(host, arrays):
                np.copyto(h.numpy(), np.asarray(a).reshape(tuple(h.shape)), casting='unsafe')
            with t.cuda.stream(self._copy_stream):
                for h, d, s8 in zip(host, devb, stage):
                    (d if s8 is None else s8).copy_(h, non_blocking=True)
                ev.record(self._copy_stream)
            self._q.put(i)
        self._q.put(None)
loop("  H2D copies only (no cast kernels on the copy stream; wrong pixels)", NoCast(Source(N), device="cuda", depth=3, consumer_stream=ad.stream))


class NoH2D(device_prefetcher):             # the reader stages into pinned memory and hands over device buffers filled once: is it the DMA?
    def _reader(self):
        t = self._torch
        for arrays in self._ds:
            i = self._slot(arrays)
            host, devb, ev, stage, _ = self._ring[i]
            for h, a in zip(host, arrays):
                np.copyto(h.numpy(), np.asarray(a).reshape(tuple(h.shape)), casting='unsafe')
            with t.cuda.stream(self._copy_stream):
                ev.record(self._copy_stream)
            self._q.put(i)
        self._q.put(None)
loop("  host staging only (no H2D, no cast; wrong pixels)", NoH2D(Source(N), device="cuda", depth=3, consumer_stream=ad.stream))


class NoStage(device_prefetcher):           # no host copy at all: only the thread hand-over
    def _reader(self):
        t = self._torch
        for arrays in self._ds:
            i = self._slot(arrays)
            host, devb, ev, stage, _ = self._ring[i]
            with t.cuda.stream(self._copy_stream):
                ev.record(self._copy_stream)
            self._q.put(i)
        self._q.put(None)
loop("  thread hand-over only (no staging, no H2D, no cast; wrong pixels)", NoStage(Source(N), device="cuda", depth=3, consumer_stream=ad.stream))
loop("prefetcher as shipped (again)", device_prefetcher(Source(N), device="cuda", depth=3, consumer_stream=ad.stream))
