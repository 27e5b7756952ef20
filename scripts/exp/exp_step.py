"""Timing experiments on the full MADNet FULL step (hipGraph replay) under the library's tuning hooks.
usage: python scripts/exp/exp_step.py [--precision bf16] [--plain-wgrad] [--wgs N]   (GPU box only)"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd"))
from madnet_hip import _ffi, engine as E, synthetic as S

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16")
ap.add_argument("--plain-wgrad", action="store_true", help="timing only: plain stores instead of atomics (wrong results)")
ap.add_argument("--wgs", type=int, default=0)
ap.add_argument("--mode", default="FULL")
ap.add_argument("--lanes", type=int, default=-1)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
dev = torch.device("cuda:0")
lib = _ffi.lib()
wn = S.calibrated_weights(dict(E.madnet_manifest()), 1)
l, r, gt = S.make_pair(375, 1242)
import numpy as np
B = a.batch
eng = E.MadNetEngine(lib, 375, 1242, B=B, device=dev, weights=wn, precision=a.precision)
eng.set_inputs(np.repeat(l, B, 0), np.repeat(r, B, 0), np.repeat(gt[..., 0], B, 0))
if a.lanes >= 0:
    eng.wgrad_lanes = a.lanes
t = a.wgs if a.wgs else (1 if a.plain_wgrad else 0)
lib.tune_wgrad_wgs(-t if a.plain_wgrad else t)        # before build_plan: the split counts are fixed when the plan is recorded
plan = eng.build_plan(a.mode, lr=1e-4)
st = torch.cuda.Stream(); sh = st.cuda_stream
with torch.cuda.stream(st):
    plan.run(lib, sh); st.synchronize()
    plan.capture(lib, sh)
    for _ in range(5): plan.launch(lib, sh)
    st.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps): plan.launch(lib, sh)
    st.synchronize()
    dt = time.perf_counter() - t0
print("precision %s mode %s plain_wgrad %s wgs %d lanes %d batch %d: %.3f ms/step = %.0f pairs/s" % (a.precision, a.mode, a.plain_wgrad, a.wgs, a.lanes, B, 1e3 * dt / a.steps, B * a.steps / dt))
