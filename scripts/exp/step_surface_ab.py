"""The reference-FPS loop (device_prefetcher -> Adapter.step, 8-bit frames) with and without the input table (mh_fetch_inputs), alternating in one process.
usage: python scripts/exp/step_surface_ab.py [steps] [rounds]"""
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

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H, W = 375, 1242
lib = _ffi.lib()
wn = S.calibrated_weights(dict(E.madnet_manifest()), 1)
pairs = [S.make_pair(H, W, stream_id=100, frame=t) for t in range(8)]
pairs8 = [(l.astype(np.uint8), r.astype(np.uint8), np.ascontiguousarray(g[..., 0])) for l, r, g in pairs]


def make(fetch):
    z = torch.zeros(1, H, W, 3, device="cuda")
    net = Nets.get_stereo_net("MADNet", {"left_img": z, "right_img": z, "split_layers": [None], "sequence": True, "train_portion": "BEGIN", "bulkhead": False, "weights": wn,
                                         "precision": "mixed", "warping": True, "context_net": True, "radius_d": 2, "stride": 1})
    ad = Adapter(net, mode="FULL", lr=1e-4, fetch_inputs=fetch)
    ad._plan("FULL")
    return ad


class Source(object):
    def __init__(self, n): self.n = n
    def __iter__(self):
        for t in range(self.n):
            yield pairs8[t % 8]


def loop(name, ad, cast, warm=20):
    k, t0 = 0, None
    for f in device_prefetcher(Source(steps + warm), device="cuda", depth=3, consumer_stream=ad.stream, cast=cast):
        if k == warm:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        ad.step(*f)
        k += 1
    torch.cuda.synchronize()
    print("%-60s %8.1f us/step" % (name, (time.perf_counter() - t0) / steps * 1e6)); sys.stdout.flush()


ads = {True: make(True), False: make(False)}
for r in range(rounds):
    for fetch in (False, True):
        for cast in (False, True):
            loop("input table %-5s  prefetcher cast=%-5s" % (fetch, cast), ads[fetch], cast)
# replay + sync alone
p = ads[False]._plan("FULL")[0]
with torch.cuda.stream(ads[False].stream):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        p.launch(lib, ads[False].stream.cuda_stream); ads[False].stream.synchronize()
print("%-60s %8.1f us/step" % ("graph replay + stream.synchronize() every step", (time.perf_counter() - t0) / steps * 1e6))
