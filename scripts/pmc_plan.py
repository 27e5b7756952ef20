"""Driver of scripts/gpu_pmc.sh (run under rocprofv3 --pmc ...): launches, one by one behind a separator that is also an L2 flush (a 96 MB fill),
  1. every conv / filter-gradient / correlation op of the recorded MADNet FULL plan ('mixed', 1242x375) -- the headline config;
  2. every such op of the MADNet MAD block plans (block_config/MadNet_piramid_only.json) whose kernel string the FULL plan did not launch -- config 3;
  3. every such op of the recorded DispNet FULL plan ('mixed') -- config 4;
  4. the fixed correlation entries of bench.py (benchtools.corr_rooflines shapes): level-2 backward plain / fused at B = 64, the 81-shift volume
     forward / backward at B = 16;
  5. (two separators in a row, then) the fixed layer entries of benchtools.roofline (3x3 128->128 forward / input gradient / streamed filter gradient, the
     estimator-2 batch, the level-2 forward correlation protocol).
Writes the launch list (group -> kernel string, algorithmic work) to $PMC_OPS_JSON so that the summariser can key the counters by the strings bench.py reports."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")):
    sys.path.insert(0, p)
import torch
from madnet_hip import _ffi, engine as E, dispnet_engine as DE, synthetic as S, benchtools as BT, ops

lib = _ffi.lib()
for item in filter(None, os.environ.get("PMC_TUNE", "").split(",")):          # PMC_TUNE="conv_bank_small=0": library tuning hooks applied before anything is recorded
    name, _, val = item.partition("=")
    getattr(lib, "tune_" + name)(int(val))
H, W = 375, 1242
l, r, gt = S.make_pair(H, W)
KINDS = (_ffi.OP_CONV, _ffi.OP_CONV_PLANES, _ffi.OP_CONV_PLANES_BWD, _ffi.OP_WGRAD_PARTIAL, _ffi.OP_WGRAD_STREAM, _ffi.OP_CORR_FWD, _ffi.OP_CORR_BWD, _ffi.OP_LEVEL_FRONT, _ffi.OP_CORR_WARP_BWD)
sep = torch.zeros(24 << 20, device="cuda")
SEP_N = sep.numel()
groups = []
seen = set()


def separator():
    lib.fill(C.c_void_p(sep.data_ptr()), SEP_N, 0.0, None)


def run_plan_ops(plan, tag, only_new=False):
    for i in range(plan.n):
        if plan.arr[i].kind not in KINDS:
            continue
        one = (_ffi.Op * 1)(plan.arr[i])
        one[0].i[26] = 0
        separator()
        lib.plan_run(one, 1, None)
        kname = BT.op_kernel_name(lib, plan.arr[i])
        fl, by = plan.work.get(i, BT.op_work(plan.arr[i]))
        groups.append({"plan": tag, "index": i, "kind": int(plan.arr[i].kind), "kernel": kname, "flops": fl, "bytes": by, "dup": kname in seen})
        seen.add(kname)
        torch.cuda.synchronize()


wn = S.calibrated_weights(dict(E.madnet_manifest()), 1)
eng = E.MadNetEngine(lib, H, W, B=1, device="cuda", weights=wn, precision="mixed")
eng.set_inputs(l, r, gt[..., 0])
plan = eng.build_plan("FULL", lr=1e-4)
plan.run(lib, 0); torch.cuda.synchronize()
run_plan_ops(plan, "madnet_full")

import Nets                                     # noqa: E402  (the MAD plans through the API surface, as bench.py's configs.mad builds them)
from madnet_hip.adapter import Adapter           # noqa: E402
tl, tr, tg = (torch.from_numpy(a).to("cuda") for a in (l, r, gt[..., 0]))
net = Nets.get_stereo_net("MADNet", {"left_img": tl, "right_img": tr, "split_layers": [None], "sequence": True, "train_portion": "BEGIN", "bulkhead": True, "weights": wn,
                                     "precision": "mixed", "warping": True, "context_net": True, "radius_d": 2, "stride": 1})
cfg = json.load(open(os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd", "block_config", "MadNet_piramid_only.json")))
ad = Adapter(net, mode="MAD", block_config=cfg, lr=1e-4, sample_mode="PROBABILITY", num_blocks=1, use_graph=False)
for b in range(len(cfg)):
    p_b = ad._plan((b,))[0]
    p_b.run(lib, 0); torch.cuda.synchronize()
    run_plan_ops(p_b, "madnet_mad_block%d" % b)

wd = S.calibrated_weights(dict(DE.dispnet_manifest()), 1)
de = DE.DispNetEngine(lib, H, W, B=1, device="cuda", weights=wd, precision="mixed")
de.set_inputs(l, r, gt[..., 0])
pd = de.build_plan("FULL", lr=1e-4)
pd.run(lib, 0); torch.cuda.synchronize()
run_plan_ops(pd, "dispnet_full")
del de, pd
torch.cuda.empty_cache()

# ---- the fixed correlation entries (one launch each behind a separator) ----------------------------------------------------------------------
fixed = {}
dev = "cuda"
B, Hc, Wc, C2, md = 64, 96, 320, 32, 2
D = 5
ld = (C2 + D + 1 + 3) // 4 * 4
L = torch.randn(B, Hc, Wc, C2, device=dev); R = torch.randn(B, Hc, Wc, C2, device=dev); g = torch.randn(B, Hc, Wc, ld, device=dev)
dL = torch.zeros_like(L); dR = torch.zeros_like(L)
gv = ops.View(g, B, Hc, Wc, ld, ld)
u = (torch.rand(B, Hc, Wc, device=dev) - 0.5) * 8.0
Rw = torch.empty_like(R); du = torch.zeros(B, Hc, Wc, device=dev)
ops.warp_fwd(lib, ops.view(R), u, ops.view(Rw)); torch.cuda.synchronize()


def fixed_entry(name, fn, byts):
    separator()
    fn()
    torch.cuda.synchronize()
    fixed[name] = lib.last_kernel().decode()
    groups.append({"plan": "fixed", "index": -1, "kind": -1, "kernel": fixed[name], "flops": 0.0, "bytes": byts, "dup": False, "fixed": name})


fixed_entry("roofline_corr_bwd", lambda: ops.corr_bwd(lib, gv, ops.view(L), ops.view(R), ops.view(dL), ops.view(dR), md, 1, coff=C2, precision=0), float(B) * Hc * Wc * (4 * C2 + D) * 4)
fixed_entry("roofline_corr_warp_bwd", lambda: ops.corr_warp_bwd(lib, gv, ops.view(L), ops.view(Rw), ops.view(R), u, ops.view(dL), ops.view(dR), du, md, 1, coff=C2, acc_l=True, copy_left=True),
            float(B) * Hc * Wc * (8 * C2 + D + 3) * 4)
del L, R, g, dL, dR, Rw, u, du
B, Cl, mdl = 16, 128, 40
D = 81
L = torch.randn(B, Hc, Wc, Cl, device=dev); R = torch.randn(B, Hc, Wc, Cl, device=dev)
vol = torch.empty(B, Hc, Wc, D, device=dev)
g = torch.randn(B, Hc, Wc, 84, device=dev); gv = ops.View(g, B, Hc, Wc, D, 84)
dL = torch.empty_like(L); dR = torch.empty_like(R)
fixed_entry("roofline_corr_d81_fwd", lambda: ops.corr_fwd(lib, ops.view(L), ops.view(R), ops.view(vol), mdl, precision=1), float(B) * Hc * Wc * (2 * Cl + D) * 4)
fixed_entry("roofline_corr_d81_bwd", lambda: ops.corr_bwd(lib, gv, ops.view(L), ops.view(R), ops.view(dL), ops.view(dR), mdl, 1, coff=0, precision=1), float(B) * Hc * Wc * (4 * Cl + D) * 4)
del L, R, vol, g, dL, dR
torch.cuda.empty_cache()

# ---- the fixed layer entries of benchtools.roofline; two separators in a row mark the boundary ------------------------------------------------------
separator(); separator()
BT.WARMUP_LAUNCHES = 1
st = torch.cuda.current_stream()
rl, extra = BT.roofline(lib, eng, st, reps=3)
torch.cuda.synchronize()
fixed["roofline_fwd"] = rl["kernel"]
for k in ("roofline_dgrad", "roofline_wgrad", "roofline_wgrad_batch", "roofline_corr"):
    if k in extra and "kernel" in extra[k]:
        fixed[k] = extra[k]["kernel"].split(" (B=")[0]
json.dump({"reps": 1, "groups": groups, "fixed": fixed}, open(os.environ.get("PMC_OPS_JSON", "/tmp/pmc_ops.json"), "w"), indent=1)
print("done: %d groups (%d distinct kernel strings)" % (len(groups), len(seen)))
