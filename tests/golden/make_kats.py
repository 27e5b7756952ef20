"""Mint the seeded known-answer fixtures of SURVEY 8(c) from the CPU oracle (oracle/tf_ops.py, oracle/madnet.py).

The reference has no tests or golden vectors and TensorFlow cannot run here ("parity unpinned"), so these
fixtures pin THIS repo's restatement: tests/test_golden_kats.py checks (a) that the oracle still reproduces
them, (b) the HIP kernels against them (CPU emulator here, MI355X under -m gpu).  Inputs are regenerated
from the seeds (torch.Generator, CPU); only the OUTPUTS are stored (float32, fp64-computed where cheap).

    python tests/golden/make_kats.py        # rewrites tests/golden/kats.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
from oracle import tf_ops as T            # noqa: E402
from oracle import madnet as OM           # noqa: E402


def rnd(shape, seed, scale=1.0, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


# name -> (B, H, W, Cin, Cout, k, stride, dilation)
CONV = {"conv_k3_s1": (1, 9, 14, 8, 12, 3, 1, 1), "conv_k3_s2": (2, 9, 13, 16, 8, 3, 2, 1), "conv_k5_s2": (1, 11, 12, 4, 8, 5, 2, 1),
        "conv_k7_s2": (1, 13, 15, 3, 8, 7, 2, 1), "conv_k3_d2": (1, 10, 12, 8, 8, 3, 1, 2), "conv_k3_d4": (1, 12, 16, 8, 4, 3, 1, 4),
        "conv_k3_d16": (1, 20, 36, 4, 4, 3, 1, 16), "conv_cout1": (1, 8, 10, 32, 1, 3, 1, 1)}
CORR = {"corr_md2": (1, 5, 24, 32, 2, 1), "corr_md2_s1_c192": (1, 3, 20, 192, 2, 1), "corr_md40": (1, 2, 70, 16, 40, 1)}


def conv_inputs(name):
    B, H, W, Ci, Co, k, s, d = CONV[name]
    sd = sum(map(ord, name))
    return rnd((B, H, W, Ci), sd), rnd((k, k, Ci, Co), sd + 1, 0.3), rnd((Co,), sd + 2), s, d


def corr_inputs(name):
    B, H, W, C, md, st = CORR[name]
    sd = sum(map(ord, name))
    return rnd((B, H, W, C), sd), rnd((B, H, W, C), sd + 1), md, st


def madnet_inputs(H=64, W=128):
    from madnet_hip import synthetic as S
    shapes = dict(OM.variable_shapes())
    wn = S.calibrated_weights(shapes, 1)
    l, r, gt = S.make_pair(H, W, stream_id=7)
    return wn, l, r, gt


def main():
    out = {}
    for name in CONV:
        x, w, b, s, d = conv_inputs(name)
        x = x.requires_grad_(True); w = w.requires_grad_(True); b = b.requires_grad_(True)
        y = T.conv2d(x, w, b, stride=s, dilation=d, alpha=0.2)
        gy = rnd(tuple(y.shape), sum(map(ord, name)) + 3)
        gx, gw, gb = torch.autograd.grad(y, [x, w, b], gy)
        out[name + "/y"], out[name + "/gx"], out[name + "/gw"], out[name + "/gb"] = y.detach(), gx, gw, gb
    # transposed conv 4x4 stride 2 (DispNet up-blocks)
    x = rnd((1, 5, 7, 8), 901); w = rnd((4, 4, 12, 8), 902, 0.3); b = rnd((12,), 903)
    out["deconv_k4_s2/y"] = T.conv2d_transpose(x, w, b, stride=2, alpha=0.1)
    for name in CORR:
        L, R, md, st = corr_inputs(name)
        L = L.requires_grad_(True); R = R.requires_grad_(True)
        c = T.correlation(L, R, md, st)
        g = rnd(tuple(c.shape), sum(map(ord, name)) + 2)
        gL, gR = torch.autograd.grad(c, [L, R], g)
        out[name + "/y"], out[name + "/gL"], out[name + "/gR"] = c.detach(), gL, gR
    # resizes (legacy TF1 bilinear, no half-pixel): x2, x64, and the 384x1280 -> crop 375x1242 form at small scale
    x = rnd((1, 6, 20, 1), 911)
    out["resize_x2/y"] = T.resize_bilinear(x, 12, 40)
    out["resize_x64/y"] = T.resize_bilinear(rnd((1, 1, 2, 1), 912), 64, 128)
    out["resize_crop/y"] = T.center_crop(T.resize_bilinear(x, 24, 80), 21, 74)
    # the two warpers, with coordinates that leave the image on both sides
    img = rnd((1, 5, 16, 8), 921); u = rnd((1, 5, 16, 1), 922, 6.0)
    img_g = img.clone().requires_grad_(True); u_g = u.clone().requires_grad_(True)
    wv = T.linear_warp(img_g, u_g)
    g = rnd(tuple(wv.shape), 923)
    gi, gu = torch.autograd.grad(wv, [img_g, u_g], g)
    out["linear_warp/y"], out["linear_warp/gimg"], out["linear_warp/gu"] = wv.detach(), gi, gu
    im3 = rnd((1, 7, 18, 3), 931).abs() * 60; d1 = rnd((1, 7, 18, 1), 932, 5.0).abs()
    out["warp_image/y"] = T.warp_image(im3, d1)
    # SSIM + L1 reprojection loss and its disparity gradient
    left = (rnd((1, 12, 24, 3), 941).abs() * 80).clamp(0, 255); right = (rnd((1, 12, 24, 3), 942).abs() * 80).clamp(0, 255)
    disp = (rnd((1, 12, 24, 1), 943, 3.0).abs()).requires_grad_(True)
    loss = T.reprojection_loss(disp, left, right)
    out["reproj_loss/loss"] = loss.detach().reshape(1)
    out["reproj_loss/gdisp"] = torch.autograd.grad(loss, disp)[0]
    # EPE / bad3
    dd = rnd((1, 9, 11, 1), 951, 4.0).abs(); gt = rnd((1, 9, 11, 1), 952, 4.0).abs(); gt[0, :2] = 0
    epe, bad = T.validation_metrics(dd, gt, 3.0)
    out["metrics/epe_bad3"] = torch.stack([torch.as_tensor(epe, dtype=torch.float64), torch.as_tensor(bad, dtype=torch.float64)])
    # momentum update
    v = rnd((1000,), 961); a = rnd((1000,), 962); gr = rnd((1000,), 963)
    acc = 0.9 * a + gr
    out["momentum/accum"], out["momentum/var"] = acc, v - 1e-4 * acc
    # offline training (Train.py): one scale of the supervised mean_l1 (gt == 0 / >= 192 invalid) with its gradient, three Adam
    # steps (fp32, the ApplyAdam functor's arithmetic)
    pr = (rnd((1, 10, 13, 1), 971, 60.0).abs()).float().requires_grad_(True)
    tg = rnd((1, 10, 13, 1), 972, 90.0).abs().float(); tg[0, 0, :3, 0] = torch.tensor([0.0, 192.0, 250.0])
    sl = T.supervised_loss(pr, tg, 0.7, 192.0)
    out["supervised_loss/loss"] = sl.detach().reshape(1).double()
    out["supervised_loss/gpred"] = torch.autograd.grad(sl, pr)[0]
    av_, am_, avv = rnd((500,), 981).float(), torch.zeros(500), torch.zeros(500)
    st = [0.9, 0.999]
    for t in range(3):
        T.adam_update(av_, am_, avv, rnd((500,), 982 + t, 0.3).float(), st, 1e-3)
        st = [float(torch.tensor(st[0], dtype=torch.float32) * torch.tensor(0.9, dtype=torch.float32)),
              float(torch.tensor(st[1], dtype=torch.float32) * torch.tensor(0.999, dtype=torch.float32))]
    out["adam/var"], out["adam/m"], out["adam/v"] = av_.clone(), am_.clone(), avv.clone()
    # MADNet: full forward (6 disparities) and one FULL step (loss, EPE, updated weights digest) at 64x128
    wn, l, r, gt = madnet_inputs()
    wt = {k: torch.from_numpy(v.copy()) for k, v in wn.items()}
    with torch.no_grad():
        disps = OM.forward(wt, torch.from_numpy(l), torch.from_numpy(r))
    for i, dsp in enumerate(disps):
        out["madnet_fwd/disp%d" % i] = dsp
    acc = {k: torch.zeros_like(v) for k, v in wt.items()}
    res = OM.step(wt, acc, torch.from_numpy(l), torch.from_numpy(r), torch.from_numpy(gt), mode="FULL", lr=1e-4)
    out["madnet_full_step/loss_epe"] = torch.tensor([float(res["loss"]), float(res["epe"])], dtype=torch.float64)
    names = sorted(wt)
    out["madnet_full_step/weight_sums"] = torch.tensor([float(wt[n].double().sum()) for n in names], dtype=torch.float64)
    out["madnet_full_step/weight_abs_sums"] = torch.tensor([float(wt[n].double().abs().sum()) for n in names], dtype=torch.float64)
    # one offline training step (multi-scale supervised loss + Adam) from the same initial weights
    wt2 = {k: torch.from_numpy(v.copy()) for k, v in wn.items()}
    m2 = {k: torch.zeros_like(v) for k, v in wt2.items()}; v2 = {k: torch.zeros_like(v) for k, v in wt2.items()}
    lw = [1.0, 0.8, 0.6, 0.4, 0.2, 0.1]
    tres = OM.train_step(wt2, m2, v2, [0.9, 0.999], torch.from_numpy(l), torch.from_numpy(r), torch.from_numpy(gt), lr=1e-4, loss_weights=lw)
    out["madnet_train_step/losses"] = torch.tensor(tres["losses"], dtype=torch.float64)
    out["madnet_train_step/weight_sums"] = torch.tensor([float(wt2[n].double().sum()) for n in names], dtype=torch.float64)
    np.savez_compressed(os.path.join(HERE, "kats.npz"), **{k: v.detach().to(torch.float32 if v.dtype != torch.float64 or v.numel() > 64 else torch.float64).numpy() for k, v in out.items()})
    print("wrote %d arrays, %.1f KiB" % (len(out), os.path.getsize(os.path.join(HERE, "kats.npz")) / 1024))


if __name__ == "__main__":
    main()
