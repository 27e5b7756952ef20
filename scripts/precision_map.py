"""Per-layer-group precision map of the MADNet forward pass at 1242x375 (CPU, oracle arithmetic): the conv operands of ONE group
at a time are rounded to bf16 (what the bf16 MFMA mode does: RNE on both operands, fp32 accumulation) or to split-bf16
(hi + lo: what precision code 2 does) while every other layer stays fp32; reported: EPE of disparities[-1] against the all-fp32
oracle.  Answers VERDICT r01 "which layers does the 0.08 px come from".  usage: python scripts/precision_map.py > profiles/r02_precision_map.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")):
    sys.path.insert(0, p)
import torch
from madnet_hip import synthetic as S
from oracle import madnet as OM
from oracle import tf_ops as T

torch.set_num_threads(min(os.cpu_count() or 1, 32))
H, W = 375, 1242
wn = S.calibrated_weights(OM.variable_shapes(), 1)
l, r, gt = S.make_pair(H, W)
wt = {k: torch.from_numpy(v.copy()) for k, v in wn.items()}
name_of = {id(v): k for k, v in wt.items()}
bf = lambda t: t.to(torch.bfloat16).to(torch.float32)


def split(t):               # hi + lo bf16: what the x3 kernels carry
    hi = bf(t)
    return hi + bf(t - hi)


orig = T.conv2d
MODE = {"groups": (), "fn": bf}


def patched(x, w, b, stride=1, dilation=1, alpha=1.0):
    n = name_of.get(id(w), "")
    if any(g in n for g in MODE["groups"]):
        # (a*b with both rounded: products of bf16 values are exact in fp32; the split form drops only lo*lo)
        return orig(MODE["fn"](x), MODE["fn"](w), b, stride, dilation, alpha)
    return orig(x, w, b, stride, dilation, alpha)


T.conv2d = patched
OM.T.conv2d = patched
with torch.no_grad():
    ref = OM.forward(wt, torch.from_numpy(l), torch.from_numpy(r))[-1]
GROUPS = [("pyramid conv1-2 (1/2 res)", ["pyramid/conv1/", "pyramid/conv2/"]), ("pyramid conv3-4 (1/4)", ["pyramid/conv3/", "pyramid/conv4/"]),
          ("pyramid conv5-12", ["pyramid/conv%d/" % i for i in range(5, 13)]), ("estimator 6", ["G6/"]), ("estimator 5", ["G5/"]),
          ("estimator 4", ["G4/"]), ("estimator 3", ["G3/"]), ("estimator 2 (1/4 res)", ["G2/"]), ("context net (1/4 res)", ["context-"]),
          ("ALL conv layers", ["/"])]
print("MADNet forward 1242x375, synthetic calibrated weights, mean |d| = %.2f px; EPE of disparities[-1] vs the all-fp32 oracle" % ref.abs().mean().item())
print("%-30s %14s %14s" % ("group rounded (others fp32)", "bf16 operands", "split-bf16"))
for name, gs in GROUPS:
    row = []
    for fn in (bf, split):
        MODE["groups"], MODE["fn"] = gs, fn
        with torch.no_grad():
            d = OM.forward(wt, torch.from_numpy(l), torch.from_numpy(r))[-1]
        row.append((d - ref).abs().mean().item())
    print("%-30s %14.3g %14.3g" % (name, row[0], row[1]))
    sys.stdout.flush()
