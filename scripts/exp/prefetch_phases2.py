"""Follow-up of prefetch_phases.py: Adapter.step is ~55 us slower behind the prefetcher even when the reader moves no data.  Is it the cross-stream event wait in front of the
graph launch, or the second Python thread?   usage: python scripts/exp/prefetch_phases2.py [steps]"""
import os, sys, time, threading, queue
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import Nets
from madnet_hip import _ffi, engine as E, synthetic as S
from madnet_hip.adapter import Adapter

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
H, W = 375, 1242
lib = _ffi.lib()
wn = S.calibrated_weights(dict(E.madnet_manifest()), 1)
pairs = [S.make_pair(H, W, stream_id=100, frame=t) for t in range(8)]
z = torch.zeros(1, H, W, 3, device="cuda")
net = Nets.get_stereo_net("MADNet", {"left_img": z, "right_img": z, "split_layers": [None], "sequence": True, "train_portion": "BEGIN", "bulkhead": False, "weights": wn,
                                     "precision": "mixed", "warping": True, "context_net": True, "radius_d": 2, "stride": 1})
ad = Adapter(net, mode="FULL", lr=1e-4)
ad._plan("FULL")
dev_frames = [tuple(torch.as_tensor(a, dtype=torch.float32, device="cuda") for a in (l, r, np.ascontiguousarray(g[..., 0]))) for l, r, g in pairs]
N, warm = steps + 20, 20


def run(name, pre=None):
    t0 = None
    for k in range(N):
        if k == warm:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        if pre: pre(k)
        ad.step(*dev_frames[k % 8])
    torch.cuda.synchronize()
    print("%-84s %8.1f us/step" % (name, (time.perf_counter() - t0) / steps * 1e6)); sys.stdout.flush()


run("frames resident, one thread, one stream")
cs = torch.cuda.Stream()
evs = [torch.cuda.Event() for _ in range(4)]
def ev_wait(k):
    e = evs[k % 4]
    e.record(cs)
    ad.stream.wait_event(e)
run("  + an event recorded on a second (idle) stream and waited on by the step's stream", ev_wait)
tiny = torch.zeros(16, device="cuda")
def ev_wait_work(k):
    e = evs[k % 4]
    with torch.cuda.stream(cs):
        tiny.add_(1.0)
    e.record(cs)
    ad.stream.wait_event(e)
run("  + a tiny kernel on the second stream before the event", ev_wait_work)
# a second Python thread that only plays queue ping-pong (no HIP calls)
qa, qb = queue.Queue(), queue.Queue()
stop = False
def pong():
    while True:
        x = qa.get()
        if x is None: return
        qb.put(x)
th = threading.Thread(target=pong, daemon=True); th.start()
def ping(k):
    qa.put(k); qb.get()
run("  one stream + a second Python thread answering a queue (no HIP call in it)", ping)
def ping_async(k):
    qa.put(k)
    try:
        qb.get_nowait()
    except queue.Empty:
        pass
run("  ... the main thread does not wait for the answer", ping_async)
qa.put(None)
# a second thread that records an event on its own stream per step (what the reader's tail does)
qc, qd = queue.Queue(), queue.Queue()
def recorder():
    k = 0
    while True:
        x = qc.get()
        if x is None: return
        e = evs[k % 4]; k += 1
        e.record(cs)
        qd.put(e)
th2 = threading.Thread(target=recorder, daemon=True); th2.start()
qc.put(0)
def handover(k):
    e = qd.get()
    ad.stream.wait_event(e)
    qc.put(k)
run("  a second thread records the event, the main thread waits on it (the prefetcher's hand-over)", handover)
qc.put(None)
run("frames resident, one thread, one stream (again)")
