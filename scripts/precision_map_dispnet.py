"""Per-layer-group precision map of the DispNet forward pass at 1242x375 (CPU, oracle arithmetic), as scripts/precision_map.py for MADNet: the conv /
deconv operands of ONE group at a time are rounded to bf16 (what precision code 1 does) while every other layer stays fp32; reported: EPE of the
final disparity against the all-fp32 oracle.  Decides which layers of the 'mixed' mode may run plain bf16 in the forward pass.
usage: python scripts/precision_map_dispnet.py > profiles/r02_precision_map_dispnet.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")):
    sys.path.insert(0, p)
import torch
from madnet_hip import synthetic as S
from oracle import dispnet as OD
from oracle import tf_ops as T

torch.set_num_threads(min(os.cpu_count() or 1, 32))
H, W = 375, 1242
wn = S.calibrated_weights(OD.variable_shapes(), 1)
l, r, gt = S.make_pair(H, W)
wt = {k: torch.from_numpy(v.copy()) for k, v in wn.items()}
name_of = {id(v): k for k, v in wt.items()}
bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
orig_c, orig_t = T.conv2d, T.conv2d_transpose
MODE = {"groups": ()}


def hit(w):
    n = name_of.get(id(w), "")
    return any(n.startswith("model/" + g + "/") or n.startswith("model/" + g + "weights") for g in MODE["groups"])


def conv(x, w, b, stride=1, dilation=1, alpha=1.0):
    return orig_c(bf(x), bf(w), b, stride, dilation, alpha) if hit(w) else orig_c(x, w, b, stride, dilation, alpha)


def deconv(x, w, b, stride=2, alpha=1.0):
    return orig_t(bf(x), bf(w), b, stride=stride, alpha=alpha) if hit(w) else orig_t(x, w, b, stride=stride, alpha=alpha)


T.conv2d = conv; T.conv2d_transpose = deconv
OD.T.conv2d = conv; OD.T.conv2d_transpose = deconv


def run():
    with torch.no_grad():
        out = OD.forward(wt, torch.from_numpy(l), torch.from_numpy(r))
    return out[-1] if isinstance(out, (list, tuple)) else out


ref = run()
names = sorted(set(k[len("model/"):].rsplit("/", 1)[0] for k in wt))
print("DispNet forward 1242x375, synthetic calibrated weights, mean |d| = %.2f px; EPE of the final disparity vs the all-fp32 oracle" % ref.abs().mean().item())
print("%-34s %14s" % ("group rounded to bf16 (others fp32)", "EPE px"))
GROUPS = [("conv1", ["conv1"]), ("conv2", ["conv2"]), ("conv_redir", ["conv_redir"]), ("conv3", ["conv3"]), ("conv3/1", ["conv3/1"]),
          ("conv4 + conv4/1", ["conv4", "conv4/1"]), ("conv5 + conv5/1", ["conv5", "conv5/1"]), ("conv6 + conv6/1", ["conv6", "conv6/1"])]
ups = sorted(set(n.split("/")[0] for n in names if n.startswith("up")))
GROUPS += [(u + " (deconv, predict, up_predict, concat)", [u + "/deconv", u + "/predict", u + "/up_predict", u + "/concat"]) for u in ups]
GROUPS += [("prediction", ["prediction"]), ("encoder conv4 .. conv6/1", ["conv4", "conv4/1", "conv5", "conv5/1", "conv6", "conv6/1"]),
           ("ALL", [n for n in names])]
for title, gs in GROUPS:
    MODE["groups"] = gs
    d = run()
    print("%-34s %14.3g" % (title, (d - ref).abs().mean().item()))
    sys.stdout.flush()
