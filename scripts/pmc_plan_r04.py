"""Driver of scripts/gpu_pmc_r04.sh: launches every conv / filter-gradient / correlation op of the recorded MADNet FULL plan ('mixed', 1242x375) one by one --
a 1-float fill launch in front of each op as a separator the summariser splits the dispatch trace at -- followed by the fixed roofline kernels of
madnet_hip/benchtools.py (the 3x3 128->128 layer forward / input gradient / streamed filter gradient, the estimator-2 batch, the correlation protocol).
Writes the op list (plan index -> kernel string) to $PMC_OPS_JSON so that the counters can be keyed by the strings bench.py reports."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")):
    sys.path.insert(0, p)
import torch
from madnet_hip import _ffi, engine as E, synthetic as S, benchtools as BT

lib = _ffi.lib()
H, W = 375, 1242
wn = S.calibrated_weights(dict(E.madnet_manifest()), 1)
l, r, gt = S.make_pair(H, W)
eng = E.MadNetEngine(lib, H, W, B=1, device="cuda", weights=wn, precision="mixed")
eng.set_inputs(l, r, gt[..., 0])
plan = eng.build_plan("FULL", lr=1e-4)
plan.run(lib, 0)
torch.cuda.synchronize()
KINDS = (_ffi.OP_CONV, _ffi.OP_CONV_PLANES, _ffi.OP_CONV_PLANES_BWD, _ffi.OP_WGRAD_PARTIAL, _ffi.OP_WGRAD_STREAM, _ffi.OP_CORR_FWD, _ffi.OP_CORR_BWD, _ffi.OP_LEVEL_FRONT, _ffi.OP_CORR_WARP_BWD)
# the separator in front of every op is ALSO an L2 flush (a 96 MB fill: three times the 32 MB of L2): round 3 measured the last of three back-to-back
# launches ("warm"), which under-reports a kernel whose operands fit the L2 of their XCD on a relaunch (the planes kernels: 31 MB in + out) -- in the
# step every layer reads what the previous kernel has just written back
sep = torch.zeros(24 << 20, device="cuda")
SEP_N = sep.numel()
ops_list = []
REPS = 1
for i in range(plan.n):
    if plan.arr[i].kind not in KINDS:
        continue
    one = (_ffi.Op * 1)(plan.arr[i])
    one[0].i[26] = 0
    lib.fill(C.c_void_p(sep.data_ptr()), SEP_N, 0.0, None)           # separator + L2 flush
    for _ in range(REPS):
        lib.plan_run(one, 1, None)
    fl, by = plan.work.get(i, BT.op_work(plan.arr[i]))
    kname = lib.last_kernel().decode()
    if plan.arr[i].kind == _ffi.OP_CONV_PLANES_BWD:
        kname = kname.replace("conv_planes_kernel<", "conv_planes_kernel<dgrad,")        # (as benchtools.plan_table labels it)
    ops_list.append({"index": i, "kind": int(plan.arr[i].kind), "kernel": kname, "flops": fl, "bytes": by})
    torch.cuda.synchronize()
# the fixed roofline kernels (names + shapes as bench.py times them); two separators in a row mark the boundary
lib.fill(C.c_void_p(sep.data_ptr()), SEP_N, 0.0, None)
lib.fill(C.c_void_p(sep.data_ptr()), SEP_N, 0.0, None)
st = torch.cuda.current_stream()
rl, extra = BT.roofline(lib, eng, st, reps=4)
torch.cuda.synchronize()
fixed = {"roofline_fwd": rl["kernel"]}
for k in ("roofline_dgrad", "roofline_wgrad", "roofline_wgrad_batch", "roofline_corr"):
    if k in extra and "kernel" in extra[k]:
        fixed[k] = extra[k]["kernel"]
json.dump({"reps": REPS, "ops": ops_list, "fixed": fixed}, open(os.environ.get("PMC_OPS_JSON", "/tmp/pmc_ops.json"), "w"), indent=1)
print("done: %d ops" % len(ops_list))
