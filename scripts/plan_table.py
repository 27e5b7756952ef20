"""Per-op table of a recorded step plan: every op timed alone (HIP events, 10 launches), kernel families by summed time.
usage: python scripts/plan_table.py [--model madnet|dispnet] [--precision mixed] [--mode FULL]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")):
    sys.path.insert(0, p)
import torch
from madnet_hip import _ffi, engine as E, dispnet_engine as DE, synthetic as S, benchtools as BT

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="madnet"); ap.add_argument("--precision", default="mixed"); ap.add_argument("--mode", default="FULL")
ap.add_argument("--rows", type=int, default=400)
ap.add_argument("--tune", action="append", default=[], help="NAME=int: calls mh_tune_NAME before the plan is recorded (conv_planes=16 ...)")
a = ap.parse_args()
lib = _ffi.lib()
for t in a.tune:
    getattr(lib, "tune_" + t.split("=")[0])(int(t.split("=")[1]))
H, W = 375, 1242
disp = a.model == "dispnet"
wn = S.calibrated_weights(dict(DE.dispnet_manifest() if disp else E.madnet_manifest()), 1)
l, r, gt = S.make_pair(H, W)
eng = (DE.DispNetEngine if disp else E.MadNetEngine)(lib, H, W, B=1, device="cuda", weights=wn, precision=a.precision)
eng.set_inputs(l, r, gt[..., 0])
plan = eng.build_plan(a.mode, lr=1e-4)
plan.run(lib, 0); torch.cuda.synchronize()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    rows, fam = BT.plan_table(lib, plan, st)
tot = sum(x[3] for x in rows)
print("%d ops, %.1f us summed stand-alone launch time" % (len(rows), tot))
for k, v in fam.items():
    print("%-70s n=%3d  %8.1f us  %5.1f %%   top %.1f us (op %d)" % (k[:70], v["launches"], v["us_per_step"], 100 * v["us_per_step"] / tot, v["top_us"], v["top_index"]))
print()
for i, kind, k, us in rows[:a.rows]:
    fl, by = plan.work.get(i, BT.op_work(plan.arr[i]))
    print("%4d kind %2d lane %d %8.1f us  %7.1f TF/s  %s" % (i, kind, plan.arr[i].i[26] & 0xff, us, fl / us * 1e-6 if us > 0 else 0, k[:110]))
