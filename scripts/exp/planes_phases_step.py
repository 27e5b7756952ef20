"""Where the plane kernels' time goes INSIDE the replayed step: the MADNet FULL step captured with parts of conv_planes_kernel switched off
(mh_tune_conv_planes timing bits: 8 = no K walk, 9 = no patch staging, 12 = no epilogue, 13 = epilogue without its stores; results are garbage, only the
clock counts).  step(all on) - step(part off) = what that part costs the step over all 33 launches of the family.
    python scripts/exp/planes_phases_step.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")):
    sys.path.insert(0, p)
import torch
from madnet_hip import _ffi, engine as E, synthetic as S, benchtools as BT

lib = _ffi.lib()
H, W = 375, 1242
wn = S.calibrated_weights(dict(E.madnet_manifest()), 1)
l, r, gt = S.make_pair(H, W)
st = torch.cuda.Stream()
sh = st.cuda_stream
MODES = [("all on", 0), ("no K walk", 1 << 8), ("no staging", 1 << 9), ("no epilogue", 1 << 12), ("epilogue w/o stores", 1 << 13), ("no walk, no epilogue", (1 << 8) | (1 << 12)),
         ("no walk, no staging, no epilogue", (1 << 8) | (1 << 9) | (1 << 12)), ("all on (again)", 0)]
base = None
for name, mode in MODES:
    lib.tune_conv_planes(mode)
    eng = E.MadNetEngine(lib, H, W, B=1, device="cuda", weights=wn, precision="mixed")
    eng.set_inputs(l, r, gt[..., 0])
    plan = eng.build_plan("FULL", lr=1e-4)
    with torch.cuda.stream(st):
        plan.run(lib, sh); st.synchronize()
        plan.capture(lib, sh)
        for _ in range(20):
            plan.launch(lib, sh)
        st.synchronize()
        ms = min(BT._time_ms(lib, st, lambda: plan.launch(lib, sh), 300) for _ in range(3))
        st.synchronize()
    if base is None:
        base = ms
    print("%-36s %8.1f us/step   %+7.1f us vs all on" % (name, ms * 1e3, (ms - base) * 1e3))
    lib.graph_destroy(plan.graph)
    del eng, plan
lib.tune_conv_planes(0)
