"""Throughput of the offline training step (Train.py defaults: crops of 320x1216, batch 4) on one MI355X: hipGraph replay
of MadNetEngine.build_plan('TRAIN') in bf16 mode, synthetic batch resident in HBM.  usage: python scripts/exp_train.py [B] [fp32]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from madnet_hip import _ffi, engine as E, synthetic as S

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
prec = "fp32" if "fp32" in sys.argv else "bf16"
lib = _ffi.lib()
H, W = 320, 1216
wn = S.calibrated_weights(dict(E.madnet_manifest()), 1)
eng = E.MadNetEngine(lib, H, W, B=B, device="cuda:0", weights=wn, precision=prec)
pairs = [S.make_pair(H, W, stream_id=i) for i in range(B)]
eng.set_inputs(np.concatenate([p[0] for p in pairs]), np.concatenate([p[1] for p in pairs]), np.concatenate([p[2][..., 0] for p in pairs]))
plan = eng.build_plan("TRAIN", lr=1e-4)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    plan.capture(lib, st.cuda_stream)
    for _ in range(3):
        plan.launch(lib, st.cuda_stream)
    st.synchronize()
    l0 = eng.res_loss_ms[:, 0].sum().item()
    t0 = time.time()
    n = 30
    for _ in range(n):
        plan.launch(lib, st.cuda_stream)
    st.synchronize()
    dt = (time.time() - t0) / n
print("TRAIN step %s B=%d %dx%d: %.3f ms/step = %.1f samples/s; ops/step %d; loss after 3 steps %.4f -> after %d steps %.4f"
      % (prec, B, H, W, dt * 1e3, B / dt, plan.n, l0, 3 + n, eng.res_loss_ms[:, 0].sum().item()))
