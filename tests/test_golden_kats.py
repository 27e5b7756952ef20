"""Known-answer fixtures (tests/golden/kats.npz, minted by tests/golden/make_kats.py from the CPU oracle).

The reference ships no golden vectors and cannot run here, so the fixtures pin this repo's restatement of it
("parity unpinned" w.r.t. the real TF1 reference -- see oracle/__init__.py / DESIGN.md):
  * test_oracle_reproduces_*  -- the oracle of today still produces the stored answers (CPU);
  * test_hip_*                -- the HIP kernels, through the C-ABI, produce them too (emulator on CPU, MI355X with -m gpu).
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_kats as MK          # noqa: E402  (input generators shared with the minting script)
from madnet_hip import ops      # noqa: E402
from oracle import tf_ops as T  # noqa: E402

KATS = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kats.npz"))


def _gold(name):
    return torch.from_numpy(KATS[name].astype(np.float64))


def _check(got, name, rtol=3e-5):
    ref = _gold(name)
    got = got.detach().cpu().double().reshape(ref.shape)
    err = (got - ref).abs().max().item()
    assert err <= rtol * max(1.0, ref.abs().max().item()), (name, err)


def _pad4(t, dev):
    c = t.shape[-1]
    ld = (c + 3) // 4 * 4
    buf = torch.zeros(*t.shape[:-1], ld, device=dev)
    buf[..., :c] = t.float().to(dev)
    return buf, ops.View(buf, t.shape[0], t.shape[1], t.shape[2], c, ld)


# ---- the oracle still reproduces the fixtures --------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(MK.CONV))
def test_oracle_reproduces_conv(name):
    x, w, b, s, d = MK.conv_inputs(name)
    _check(T.conv2d(x, w, b, stride=s, dilation=d, alpha=0.2), name + "/y", 1e-6)


@pytest.mark.parametrize("name", sorted(MK.CORR))
def test_oracle_reproduces_corr(name):
    L, R, md, st = MK.corr_inputs(name)
    _check(T.correlation(L, R, md, st), name + "/y", 1e-6)


def test_oracle_reproduces_resize_and_warp():
    x = MK.rnd((1, 6, 20, 1), 911)
    _check(T.resize_bilinear(x, 12, 40), "resize_x2/y", 1e-6)
    _check(T.center_crop(T.resize_bilinear(x, 24, 80), 21, 74), "resize_crop/y", 1e-6)
    _check(T.linear_warp(MK.rnd((1, 5, 16, 8), 921), MK.rnd((1, 5, 16, 1), 922, 6.0)), "linear_warp/y", 1e-6)


def test_oracle_reproduces_madnet_forward():
    wn, l, r, gt = MK.madnet_inputs()
    wt = {k: torch.from_numpy(v.copy()) for k, v in wn.items()}
    from oracle import madnet as OM
    with torch.no_grad():
        disps = OM.forward(wt, torch.from_numpy(l), torch.from_numpy(r))
    assert len(disps) == 6                              # Stereo_net.get_disparities(): [d6,d5,d4,d3,d2ctx,full]
    for i, d in enumerate(disps):
        _check(d, "madnet_fwd/disp%d" % i, 2e-5)


# ---- the HIP kernels reproduce the fixtures ------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(MK.CONV))
def test_hip_conv_kats(backend, name):
    dev = backend.device
    x, w, b, s, d = MK.conv_inputs(name)
    B, H, W, Ci = x.shape
    k, Co = w.shape[0], w.shape[3]
    Ho, Wo, _, _ = ops.conv_geometry(H, W, k, k, s, d)
    xb, xv = _pad4(x, dev)
    wf, bf = w.float().to(dev).contiguous(), b.float().to(dev)
    y = torch.full((B, Ho, Wo, Co), float("nan"), device=dev)
    ops.conv2d_fwd(backend.lib, xv, wf, bf, ops.view(y), stride=s, dil=d, alpha=0.2)
    backend.sync()
    _check(y, name + "/y")
    # gradients of the leaky output w.r.t. x, w, b given gy:  dz = gy * leaky'(y)
    gy = MK.rnd((B, Ho, Wo, Co), sum(map(ord, name)) + 3).float().to(dev)
    dz = (gy * torch.where(y > 0, 1.0, 0.2)).contiguous()
    zb, zv = _pad4(dz.cpu(), dev)
    dxb, dxv = _pad4(torch.zeros(B, H, W, Ci), dev)
    if s in (1, 2):
        ops.conv2d_dgrad(backend.lib, zv, wf, dxv, stride=s, dil=d)
    dw = torch.zeros_like(wf); db = torch.zeros_like(bf)
    ops.conv2d_wgrad(backend.lib, xv, zv, dw, db, stride=s, dil=d)
    backend.sync()
    _check(dxb[..., :Ci], name + "/gx", 1e-4)
    _check(dw, name + "/gw", 1e-4)
    _check(db, name + "/gb", 1e-4)


def test_hip_deconv_kat(backend):
    dev = backend.device
    x = MK.rnd((1, 5, 7, 8), 901).float().to(dev); w = MK.rnd((4, 4, 12, 8), 902, 0.3).float().to(dev).contiguous()
    b = MK.rnd((12,), 903).float().to(dev)
    y = torch.full((1, 10, 14, 12), float("nan"), device=dev)
    ops.conv2d_transpose_fwd(backend.lib, ops.view(x), w, b, ops.view(y), stride=2, alpha=0.1)
    backend.sync()
    _check(y, "deconv_k4_s2/y")


@pytest.mark.parametrize("name", sorted(MK.CORR))
def test_hip_corr_kats(backend, name):
    dev = backend.device
    L, R, md, st = MK.corr_inputs(name)
    B, H, W, C = L.shape
    Lf, Rf = L.float().to(dev), R.float().to(dev)
    D = 2 * md // st + 1
    out = torch.full((B, H, W, D), float("nan"), device=dev)
    ops.corr_fwd(backend.lib, ops.view(Lf), ops.view(Rf), ops.view(out), md, st)
    g = MK.rnd((B, H, W, D), sum(map(ord, name)) + 2).float().to(dev)
    dL = torch.full_like(Lf, float("nan")); dR = torch.full_like(Rf, float("nan"))
    ops.corr_bwd(backend.lib, ops.view(g), ops.view(Lf), ops.view(Rf), ops.view(dL), ops.view(dR), md, st)
    backend.sync()
    _check(out, name + "/y")
    _check(dL, name + "/gL", 1e-4)
    _check(dR, name + "/gR", 1e-4)


def test_hip_resize_warp_kats(backend):
    dev = backend.device
    x = MK.rnd((1, 6, 20, 1), 911).float().to(dev)[..., 0].contiguous()
    o = torch.full((1, 12, 40), float("nan"), device=dev)
    ops.resize_fwd(backend.lib, x, o, 12, 40)
    o2 = torch.full((1, 21, 74), float("nan"), device=dev)
    ops.resize_fwd(backend.lib, x, o2, 24, 80, cy=(24 - 21) // 2, cx=(80 - 74) // 2)
    x64 = MK.rnd((1, 1, 2, 1), 912).float().to(dev)[..., 0].contiguous()
    o3 = torch.full((1, 64, 128), float("nan"), device=dev)
    ops.resize_fwd(backend.lib, x64, o3, 64, 128)
    img = MK.rnd((1, 5, 16, 8), 921).float().to(dev); u = MK.rnd((1, 5, 16, 1), 922, 6.0).float().to(dev)[..., 0].contiguous()
    wv = torch.full_like(img, float("nan"))
    ops.warp_fwd(backend.lib, ops.view(img), u, ops.view(wv))
    g = MK.rnd((1, 5, 16, 8), 923).float().to(dev)
    dimg = torch.zeros_like(img); du = torch.zeros_like(u)
    ops.warp_bwd(backend.lib, ops.view(g), ops.view(img), u, ops.view(dimg), du=du, acc_u=True)
    backend.sync()
    _check(o, "resize_x2/y"); _check(o2, "resize_crop/y"); _check(o3, "resize_x64/y")
    _check(wv, "linear_warp/y"); _check(dimg, "linear_warp/gimg", 1e-4); _check(du, "linear_warp/gu", 1e-4)


def test_hip_loss_metrics_momentum_kats(backend):
    dev = backend.device
    left = (MK.rnd((1, 12, 24, 3), 941).abs() * 80).clamp(0, 255).float().to(dev)
    right = (MK.rnd((1, 12, 24, 3), 942).abs() * 80).clamp(0, 255).float().to(dev)
    disp = MK.rnd((1, 12, 24, 1), 943, 3.0).abs().float().to(dev)[..., 0].contiguous()
    ws = torch.zeros(backend.lib.loss_ws_floats(1, 12, 24), device=dev)
    res = torch.zeros(4, device=dev); dd = torch.full((1, 12, 24), float("nan"), device=dev)
    ops.reprojection_loss(backend.lib, left, right, disp, ws, res, dd)
    d = MK.rnd((1, 9, 11, 1), 951, 4.0).abs().float().to(dev)[..., 0].contiguous()
    gt = MK.rnd((1, 9, 11, 1), 952, 4.0).abs(); gt[0, :2] = 0
    gt = gt.float().to(dev)[..., 0].contiguous()
    mws = torch.zeros(backend.lib.metrics_ws_floats(1, 9, 11), device=dev); mres = torch.zeros(4, device=dev)
    ops.metrics(backend.lib, d, gt, mws, mres, 3.0)
    v = MK.rnd((1000,), 961).float().to(dev); a = MK.rnd((1000,), 962).float().to(dev); gr = MK.rnd((1000,), 963).float().to(dev)
    ops.momentum(backend.lib, v, a, gr, lr=1e-4, mom=0.9)
    backend.sync()
    _check(res[:1], "reproj_loss/loss", 5e-6)
    _check(dd, "reproj_loss/gdisp", 5e-4)
    _check(mres[:2], "metrics/epe_bad3", 1e-5)
    _check(a, "momentum/accum", 1e-6); _check(v, "momentum/var", 1e-6)


def test_hip_madnet_forward_and_full_step_kats(backend):
    """6 disparities of one forward + loss / EPE / per-variable weight digests after one FULL step, 64x128."""
    from madnet_hip import engine as E
    if backend.name == "emul":
        pytest.skip("covered at this size by tests/test_engine_parity.py on the emulator (minutes per run)")
    dev = backend.device
    wn, l, r, gt = MK.madnet_inputs()
    eng = E.MadNetEngine(backend.lib, 64, 128, B=1, device=dev, weights=wn)
    eng.set_inputs(l, r, gt[..., 0])
    from madnet_hip.plan import Recorder
    rec = Recorder()
    eng.record_forward(rec, make_disps=tuple(E.LEVELS))          # forward + the _make_disp of every level
    rec.compile().run(backend.lib, 0)
    backend.sync()
    order = [6, 5, 4, 3, 2]
    for i, k in enumerate(order):
        _check(eng.disp_k[k], "madnet_fwd/disp%d" % i, 1e-4)
    _check(eng.pred, "madnet_fwd/disp5", 1e-4)
    eng.build_plan("FULL", lr=1e-4).run(backend.lib, 0)
    backend.sync()
    le = _gold("madnet_full_step/loss_epe")
    assert abs(eng.res_loss[0].item() - le[0].item()) <= 1e-5 * max(1.0, abs(le[0].item()))
    assert abs(eng.res_met[0].item() - le[1].item()) <= 1e-4 * max(1.0, abs(le[1].item()))
    names = sorted(wn)
    sums = torch.tensor([float(eng.params.tensor(n).double().sum()) for n in names], dtype=torch.float64)
    asums = torch.tensor([float(eng.params.tensor(n).double().abs().sum()) for n in names], dtype=torch.float64)
    ref_s, ref_a = _gold("madnet_full_step/weight_sums"), _gold("madnet_full_step/weight_abs_sums")
    assert ((sums - ref_s).abs() <= 2e-5 * ref_a.clamp(min=1.0)).all()
    assert ((asums - ref_a).abs() <= 2e-5 * ref_a.clamp(min=1.0)).all()


def _train_kat_inputs():
    pr = MK.rnd((1, 10, 13, 1), 971, 60.0).abs().float()
    tg = MK.rnd((1, 10, 13, 1), 972, 90.0).abs().float(); tg[0, 0, :3, 0] = torch.tensor([0.0, 192.0, 250.0])
    return pr, tg


def test_oracle_reproduces_training_kats():
    pr, tg = _train_kat_inputs()
    pr = pr.requires_grad_(True)
    sl = T.supervised_loss(pr, tg, 0.7, 192.0)
    _check(sl.reshape(1), "supervised_loss/loss", 1e-6)
    _check(torch.autograd.grad(sl, pr)[0], "supervised_loss/gpred", 1e-6)
    v, m, vv, st = MK.rnd((500,), 981).float(), torch.zeros(500), torch.zeros(500), [0.9, 0.999]
    for t in range(3):
        T.adam_update(v, m, vv, MK.rnd((500,), 982 + t, 0.3).float(), st, 1e-3)
        st = [float(torch.tensor(st[0], dtype=torch.float32) * torch.tensor(0.9, dtype=torch.float32)),
              float(torch.tensor(st[1], dtype=torch.float32) * torch.tensor(0.999, dtype=torch.float32))]
    _check(v, "adam/var", 1e-7); _check(m, "adam/m", 1e-7); _check(vv, "adam/v", 1e-7)


def test_hip_training_kats(backend):
    """Train.py's kernels against the stored answers: supervised mean_l1 (+ gradient) and three Adam steps."""
    dev = backend.device
    pr, tg = _train_kat_inputs()
    pred = pr[..., 0].contiguous().to(dev); gt = tg[..., 0].contiguous().to(dev)
    ws = torch.zeros(backend.lib.proxy_ws_floats(1, 10, 13), device=dev); res = torch.zeros(4, device=dev)
    dp = torch.full((1, 10, 13), float("nan"), device=dev)
    ops.supervised_loss(backend.lib, pred, gt, ws, res, dp, weight=0.7, max_disp=192.0)
    v = MK.rnd((500,), 981).float().to(dev); m = torch.zeros(500, device=dev); vv = torch.zeros(500, device=dev)
    st = torch.tensor([0.9, 0.999], device=dev)
    for t in range(3):
        ops.adam(backend.lib, v, m, vv, MK.rnd((500,), 982 + t, 0.3).float().to(dev), st, lr=1e-3)
        ops.adam_advance(backend.lib, st)
    backend.sync()
    _check(res[:1], "supervised_loss/loss", 2e-6)
    _check(dp, "supervised_loss/gpred", 1e-5)
    _check(v, "adam/var", 2e-6); _check(m, "adam/m", 2e-6); _check(vv, "adam/v", 2e-6)


@pytest.mark.gpu
def test_hip_madnet_train_step_kat(hip):
    """The six loss terms and per-variable weight digests after one offline training step at 64x128 (lr 1e-4)."""
    from madnet_hip import engine as E
    wn, l, r, gt = MK.madnet_inputs()
    eng = E.MadNetEngine(hip.lib, 64, 128, B=1, device=hip.device, weights=wn)
    eng.set_inputs(l, r, gt[..., 0])
    eng.build_plan("TRAIN", lr=1e-4, loss_weights=[1.0, 0.8, 0.6, 0.4, 0.2, 0.1]).run(hip.lib, 0)
    hip.sync()
    ref = _gold("madnet_train_step/losses")
    got = eng.res_loss_ms[:, 0].cpu().double()
    assert ((got - ref).abs() <= 5e-5 * ref.abs().clamp(min=1.0)).all(), (got, ref)
    names = sorted(wn)
    sums = torch.tensor([float(eng.params.tensor(n).double().sum()) for n in names], dtype=torch.float64)
    ref_s, ref_a = _gold("madnet_train_step/weight_sums"), _gold("madnet_full_step/weight_abs_sums")
    # every element moves by ~lr: the digest of a tensor may differ by a small fraction of (lr * numel)
    numel = torch.tensor([float(eng.params.numel(n)) for n in names], dtype=torch.float64)
    assert ((sums - ref_s).abs() <= 2e-5 * ref_a.clamp(min=1.0) + 2e-3 * 1e-4 * numel).all()

