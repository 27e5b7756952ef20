"""Correlation / warp / resize / pad / loss / metrics / momentum parity vs the torch oracle
(same tests run on the CPU-emulated build and, marked gpu, on the MI355X)."""
import numpy as np
import pytest
import torch

from madnet_hip import ops
from oracle import tf_ops as T


def _rand(shape, seed, dev, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def _close(a, b, rtol=2e-5, atol=2e-6):
    a, b = a.detach().cpu(), b.detach().cpu()
    err = (a - b).abs().max().item()
    return err <= atol + rtol * max(1.0, b.abs().max().item()), err


CORR_CASES = [(1, 6, 20, 192, 2, 1), (1, 5, 70, 32, 2, 1), (2, 4, 9, 64, 2, 1), (1, 3, 33, 96, 3, 2),
              (1, 3, 17, 16, 4, 1), (1, 2, 40, 32, 10, 1), (1, 2, 70, 128, 40, 1)]


@pytest.mark.parametrize("case", CORR_CASES)
def test_corr_fwd_concat(backend, case):
    """[L | corr | u | 0-pad] written by one kernel == tf.concat of the oracle pieces."""
    B, H, W, C, md, st = case
    if backend.name == "emul" and md == 40:
        H = 1
    dev = backend.device
    L = _rand((B, H, W, C), 1, dev); R = _rand((B, H, W, C), 2, dev); u = _rand((B, H, W), 3, dev)
    ref = T.correlation(L.cpu(), R.cpu(), md, st)
    D = ref.shape[-1]
    ld = (C + D + 1 + 3) // 4 * 4
    out = torch.full((B, H, W, ld), float("nan"), device=dev)
    ov = ops.View(out, B, H, W, ld, ld)
    ops.corr_fwd(backend.lib, ops.view(L), ops.view(R), ov, md, st, coff=C, u=u, copy_left=True, zero_tail=True)
    backend.sync()
    o = out.cpu()
    assert torch.equal(o[..., :C], L.cpu())
    ok, err = _close(o[..., C:C + D], ref); assert ok, err
    assert torch.equal(o[..., C + D], u.cpu())
    assert torch.all(o[..., C + D + 1:] == 0)
    # stand-alone form (sharedLayers.correlation): corr only, no concat
    out2 = torch.full((B, H, W, D), float("nan"), device=dev)
    ops.corr_fwd(backend.lib, ops.view(L), ops.view(R), ops.view(out2), md, st)
    backend.sync()
    ok, err = _close(out2, ref); assert ok, err


@pytest.mark.parametrize("case", CORR_CASES[:6])
def test_corr_bwd(backend, case):
    B, H, W, C, md, st = case
    dev = backend.device
    L = _rand((B, H, W, C), 4, dev); R = _rand((B, H, W, C), 5, dev)
    Lc = L.cpu().requires_grad_(True); Rc = R.cpu().requires_grad_(True)
    ref = T.correlation(Lc, Rc, md, st)
    D = ref.shape[-1]
    ld = (C + D + 1 + 3) // 4 * 4
    g = _rand((B, H, W, ld), 6, dev)
    gl_ref, gr_ref = torch.autograd.grad(ref, [Lc, Rc], g.cpu()[..., C:C + D])
    old_l = _rand((B, H, W, C), 7, dev)
    dL = old_l.clone(); dR = torch.full((B, H, W, C), float("nan"), device=dev); du = torch.zeros(B, H, W, device=dev)
    gv = ops.View(g, B, H, W, ld, ld)
    ops.corr_bwd(backend.lib, gv, ops.view(L), ops.view(R), ops.view(dL), ops.view(dR), md, st, coff=C, du=du,
                 acc_l=True, acc_r=False, acc_u=False, copy_left=True)
    backend.sync()
    ok, err = _close(dL, old_l.cpu() + g.cpu()[..., :C] + gl_ref); assert ok, err
    ok, err = _close(dR, gr_ref); assert ok, err
    assert torch.equal(du.cpu(), g.cpu()[..., C + D])


@pytest.mark.parametrize("case", [(1, 1, 70, 16, 40, 0, 0, 2), (2, 2, 37, 32, 40, 1, 0, 0), (1, 1, 130, 128, 40, 0, 1, 0), (1, 1, 64, 16, 20, 1, 1, 4),
                                  (1, 2, 37, 48, 40, 0, 0, 0), (1, 1, 40, 64, 40, 0, 0, 1)])
def test_corr_bwd_standalone_large_shift(backend, case):
    """DispNet form (sharedLayers.correlation alone, D = 2*md+1 > 9): the banded-GEMM MFMA gradient kernels
    (C a power of two; 16-byte and unaligned g rows) and the gather fallback (C = 48) against autograd of the oracle, with/without accumulation."""
    B, H, W, C, md, al, ar, coff = case
    dev = backend.device
    L = _rand((B, H, W, C), 41, dev); R = _rand((B, H, W, C), 42, dev)
    Lc = L.cpu().requires_grad_(True); Rc = R.cpu().requires_grad_(True)
    ref = T.correlation(Lc, Rc, md, 1)
    D = ref.shape[-1]
    ld = (D + 3) // 4 * 4 + 4
    g = _rand((B, H, W, ld), 43, dev)
    gl_ref, gr_ref = torch.autograd.grad(ref, [Lc, Rc], g.cpu()[..., coff:coff + D])
    old_l = _rand((B, H, W, C), 44, dev); old_r = _rand((B, H, W, C), 45, dev)
    dL = old_l.clone() if al else torch.full((B, H, W, C), float("nan"), device=dev)
    dR = old_r.clone() if ar else torch.full((B, H, W, C), float("nan"), device=dev)
    ops.corr_bwd(backend.lib, ops.View(g, B, H, W, ld, ld), ops.view(L), ops.view(R), ops.view(dL), ops.view(dR), md, 1, coff=coff,
                 acc_l=bool(al), acc_r=bool(ar), copy_left=False)
    backend.sync()
    ok, err = _close(dL, gl_ref + (old_l.cpu() if al else 0)); assert ok, err
    ok, err = _close(dR, gr_ref + (old_r.cpu() if ar else 0)); assert ok, err


@pytest.mark.parametrize("case", [(1, 6, 20, 128), (2, 5, 33, 32), (1, 4, 17, 16), (1, 3, 40, 96)])
def test_warp_fwd_bwd(backend, case):
    B, H, W, C = case
    dev = backend.device
    img = _rand((B, H, W, C), 8, dev)
    u = _rand((B, H, W, 1), 9, dev, 4.0)
    u[0, 0, 0, 0] = -3.0; u[0, 0, W - 1, 0] = 2.5; u[0, 1, 2, 0] = 1.0; u[0, 1, 3, 0] = -100.0   # out of range / integral
    ic = img.cpu().requires_grad_(True); uc = u.cpu().requires_grad_(True)
    ref = T.linear_warp(ic, uc)
    g = _rand((B, H, W, C), 10, dev)
    gi_ref, gu_ref = torch.autograd.grad(ref, [ic, uc], g.cpu())
    out = torch.full((B, H, W, C), float("nan"), device=dev)
    u3 = u[..., 0].contiguous()
    ops.warp_fwd(backend.lib, ops.view(img), u3, ops.view(out))
    dimg = torch.zeros(B, H, W, C, device=dev); du = torch.full((B, H, W), 0.5, device=dev)
    ops.warp_bwd(backend.lib, ops.view(g), ops.view(img), u3, ops.view(dimg), du=du, acc_u=True)
    backend.sync()
    ok, err = _close(out, ref); assert ok, err
    ok, err = _close(dimg, gi_ref, rtol=5e-5); assert ok, err
    ok, err = _close(du, gu_ref[..., 0] + 0.5, rtol=5e-5, atol=2e-5); assert ok, err


RESIZE_CASES = [  # Hi, Wi, Hr, Wr, cy, cx, Ho, Wo, mul, mode
    (6, 20, 12, 40, 0, 0, 12, 40, 20.0 / 32, 0),       # inter-level upsample x2
    (6, 20, 384, 1280, 4, 19, 375, 1242, -20.0, 1),    # _make_disp from level 6 (x64), relu first
    (24, 80, 96, 320, 1, 5, 93, 311, -20.0, 2),        # final prediction style (x4), relu after
    (7, 9, 7, 9, 0, 0, 7, 9, 1.0, 0),                  # identity size
    (9, 11, 5, 7, 0, 1, 5, 5, 0.5, 0),                 # down-scaling, non-integer ratio
    (5, 8, 13, 21, 2, 3, 9, 15, -1.5, 2),              # non-integer up ratio
]


@pytest.mark.parametrize("case", RESIZE_CASES)
def test_resize_fwd_bwd(backend, case):
    Hi, Wi, Hr, Wr, cy, cx, Ho, Wo, mul, mode = case
    dev = backend.device
    B = 2
    x = _rand((B, Hi, Wi), 11, dev)
    xc = x.cpu().requires_grad_(True)
    xin = xc[..., None]
    if mode == 0:
        r = T.resize_bilinear(xin, Hr, Wr) * mul
    elif mode == 1:
        r = T.resize_bilinear(torch.relu(xin * mul), Hr, Wr)
    else:
        r = torch.relu(T.resize_bilinear(xin, Hr, Wr) * mul)
    ref = r[:, cy:cy + Ho, cx:cx + Wo, 0]
    g = _rand((B, Ho, Wo), 12, dev)
    (gx_ref,) = torch.autograd.grad(ref, [xc], g.cpu())
    out = torch.full((B, Ho, Wo), float("nan"), device=dev)
    ops.resize_fwd(backend.lib, x, out, Hr, Wr, cy, cx, mul, mode)
    old = _rand((B, Hi, Wi), 13, dev)
    dx = old.clone()
    ops.resize_bwd(backend.lib, g, x, dx, Hr, Wr, cy, cx, mul, mode, accumulate=True)
    backend.sync()
    ok, err = _close(out, ref); assert ok, err
    ok, err = _close(dx, old.cpu() + gx_ref, rtol=5e-5, atol=1e-5); assert ok, err


def test_pad_reflect(backend):
    dev = backend.device
    x = torch.arange(2 * 11 * 14 * 3, dtype=torch.float32).reshape(2, 11, 14, 3).to(dev)
    ref = T.pad_image(x.cpu(), 8)
    Hp, Wp = ref.shape[1], ref.shape[2]
    out = torch.full((2, Hp, Wp, 4), float("nan"), device=dev)
    ops.pad_reflect(backend.lib, x, out, (Hp - 11) // 2, (Wp - 14) // 2)
    backend.sync()
    assert torch.equal(out.cpu()[..., :3], ref)          # index path: bit exact
    assert torch.all(out.cpu()[..., 3] == 0)


@pytest.mark.parametrize("case", [(2, 11, 14, 8, 2, 1.0, 0.0), (2, 41, 37, 64, 2, 1.0, 0.0), (1, 30, 23, 16, 1, 1.0, 0.0), (2, 13, 18, 8, 2, 255.0, 100.0 / 255), (4, 60, 100, 64, 2, 1.0, 0.0)])
def test_conv_image_fwd_from_the_frames(backend, case):
    """mh_conv_image_fwd (conv1 straight from the frames through the reflection, MadNet.py:56-60 after Stereo_net._preprocess_inputs -> pad_image) == the oracle's
    pad_image + conv2d(stride) + leaky, and == mh_pad_reflect + mh_conv2d_fwd (exact fp32) -- padded sizes that are odd / even (the SAME padding of a stride-2
    layer sits at the END of an even frame), reflection depths up to the frame size - 1, the x / div - sub preprocessing, the bf16 shadow, NaN canaries."""
    NB, H0, W0, factor, stride, div, sub = case
    dev, lib = backend.device, backend.lib
    g = torch.Generator().manual_seed(11)
    frames = torch.floor(torch.rand(NB, H0, W0, 3, generator=g) * 256).to(dev)
    w = _rand((3, 3, 3, 16), 12, dev, 0.3); b = _rand((16,), 13, dev)
    xp = T.pad_image((frames.cpu() / div - sub) if (div != 1.0 or sub != 0.0) else frames.cpu(), factor)
    Hp, Wp = xp.shape[1], xp.shape[2]
    pt, pl = (Hp - H0) // 2, (Wp - W0) // 2
    ref = T.conv2d(xp, w.cpu(), b.cpu(), stride, 1, 0.2)
    Ho, Wo = ref.shape[1], ref.shape[2]
    out = torch.full((NB, Ho, Wo, 16), float("nan"), device=dev)
    sh = ops.Shadow(NB, Ho, Wo, 16, dev)
    assert lib.conv_image_ok(3, 16, 3, 3, stride) == 1 and lib.conv_image_ok(4, 16, 3, 3, 2) == 0 and lib.conv_image_ok(3, 32, 3, 3, 2) == 0
    ops.conv_image_fwd(lib, frames, Hp, Wp, pt, pl, w, b, ops.view(out), stride=stride, alpha=0.2, div=div, sub=sub, shadow=sh)
    assert "conv_image_fwd_kernel" in lib.last_kernel().decode()
    # the two launches it replaces, exact fp32
    x0 = torch.zeros(NB, Hp, Wp, 4, device=dev)
    ops.pad_reflect(lib, frames, x0, pt, pl, div=div, sub=sub)
    out2 = torch.full((NB, Ho, Wo, 16), float("nan"), device=dev)
    ops.conv2d_fwd(lib, ops.View(x0, NB, Hp, Wp, 3, 4), w, b, ops.view(out2), stride=stride, alpha=0.2, precision=0)
    backend.sync()
    o = out.cpu()
    assert torch.isfinite(o).all()
    scale = max(1.0, ref.abs().max().item())
    assert (o - ref).abs().max().item() <= 2e-6 * scale, (o - ref).abs().max().item()
    assert (o - out2.cpu()).abs().max().item() <= 2e-6 * scale
    shv = sh.t.float().cpu()
    assert torch.equal(shv[..., :16], o.to(torch.bfloat16).float()) and torch.all(shv[..., 16:] == 0)


@pytest.mark.parametrize("shape", [(1, 9, 14), (2, 12, 21)])
def test_reprojection_loss_and_grad(backend, shape):
    B, H, W = shape
    dev = backend.device
    g = torch.Generator().manual_seed(5)
    left = torch.floor(torch.rand(B, H, W, 3, generator=g) * 256).to(dev)
    right = torch.floor(torch.rand(B, H, W, 3, generator=g) * 256).to(dev)
    disp = (torch.rand(B, H, W, generator=g) * 6 - 1).to(dev)       # includes negative / out of range warps
    dc = disp.cpu().requires_grad_(True)
    ref = T.reprojection_loss(dc[..., None], left.cpu(), right.cpu())
    (gd_ref,) = torch.autograd.grad(ref, [dc])
    ws = torch.zeros(backend.lib.loss_ws_floats(B, H, W), device=dev)
    res = torch.zeros(4, device=dev); dd = torch.full((B, H, W), float("nan"), device=dev)
    ops.reprojection_loss(backend.lib, left, right, disp, ws, res, dd, grad_scale=1.0)
    backend.sync()
    assert abs(res[0].item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item()))
    ok, err = _close(dd, gd_ref, rtol=2e-4, atol=1e-7); assert ok, err


def test_reprojection_loss_smooth_images(backend):
    """Low-texture images: SSIM denominators near C1/C2 and clip boundaries are exercised."""
    B, H, W = 1, 10, 16
    dev = backend.device
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    left = torch.stack([xx * 3 + 10, yy * 5 + 20, xx + yy], -1)[None].to(dev)
    right = torch.stack([xx * 3 + 16, yy * 5 + 20, xx + yy + 2], -1)[None].to(dev)
    disp = torch.full((B, H, W), 2.25).to(dev)
    dc = disp.cpu().requires_grad_(True)
    ref = T.reprojection_loss(dc[..., None], left.cpu(), right.cpu())
    (gd_ref,) = torch.autograd.grad(ref, [dc])
    ws = torch.zeros(backend.lib.loss_ws_floats(B, H, W), device=dev)
    res = torch.zeros(4, device=dev); dd = torch.zeros(B, H, W, device=dev)
    ops.reprojection_loss(backend.lib, left, right, disp, ws, res, dd)
    backend.sync()
    assert abs(res[0].item() - ref.item()) <= 5e-6
    ok, err = _close(dd, gd_ref, rtol=1e-3, atol=1e-6); assert ok, err


def test_metrics(backend):
    B, H, W = 1, 13, 29
    dev = backend.device
    g = torch.Generator().manual_seed(9)
    disp = (torch.rand(B, H, W, generator=g) * 50).to(dev)
    gt = (torch.rand(B, H, W, generator=g) * 50)
    gt[torch.rand(B, H, W, generator=g) < 0.7] = 0.0
    gt = gt.to(dev)
    epe, bad = T.validation_metrics(disp.cpu()[..., None], gt.cpu()[..., None])
    ws = torch.zeros(backend.lib.metrics_ws_floats(B, H, W), device=dev); res = torch.zeros(4, device=dev)
    ops.metrics(backend.lib, disp, gt, ws, res, 3.0)
    backend.sync()
    assert abs(res[0].item() - epe.item()) <= 1e-5 * epe.item()
    assert abs(res[1].item() - bad.item()) <= 1e-6
    assert res[2].item() == float((gt != 0).sum())


def test_momentum_and_glue(backend):
    dev = backend.device
    n = 1000
    w = _rand((n,), 1, dev); m = _rand((n,), 2, dev); g = _rand((n,), 3, dev)
    w0, m0 = w.cpu().clone(), m.cpu().clone()
    ops.momentum(backend.lib, w, m, g, lr=1e-2, mom=0.9, grad_scale=0.5)
    backend.sync()
    m_ref = 0.9 * m0 + 0.5 * g.cpu()
    assert torch.allclose(m.cpu(), m_ref, rtol=1e-6, atol=1e-7)
    assert torch.allclose(w.cpu(), w0 - 1e-2 * m_ref, rtol=1e-6, atol=1e-7)
    # copy_channels into a slice of a wider buffer, accumulate + scale
    src = _rand((1, 4, 5, 3), 4, dev); dst = _rand((1, 4, 5, 8), 5, dev); d0 = dst.cpu().clone()
    dv = ops.View(dst, 1, 4, 5, 3, 8, coff=2)
    ops.copy_channels(backend.lib, ops.view(src), dv, scale=2.0, accumulate=True)
    y = _rand((1, 4, 5, 8), 6, dev); dy = _rand((1, 4, 5, 8), 7, dev); dy0 = dy.cpu().clone()
    ops.leaky_bwd(backend.lib, ops.view(dy), ops.view(y), 0.2)
    backend.sync()
    exp = d0.clone(); exp[..., 2:5] += 2.0 * src.cpu()
    assert torch.allclose(dst.cpu(), exp)
    assert torch.allclose(dy.cpu(), dy0 * torch.where(y.cpu() > 0, 1.0, 0.2))


@pytest.mark.parametrize("direct", [0, 1])
def test_corr_fwd_both_kernels(backend, direct):
    """LDS-staged window kernel vs direct (bounds-checked buffer loads) kernel: same values."""
    B, H, W, C, md = 2, 3, 37, 32, 2
    dev = backend.device
    L = _rand((B, H, W, C), 21, dev); R = _rand((B, H, W, C), 22, dev)
    ref = T.correlation(L.cpu(), R.cpu(), md, 1)
    out = torch.full((B, H, W, 5), float("nan"), device=dev)
    backend.lib.tune_corr(direct)
    try:
        ops.corr_fwd(backend.lib, ops.view(L), ops.view(R), ops.view(out), md, 1)
        backend.sync()
    finally:
        backend.lib.tune_corr(1)
    ok, err = _close(out, ref); assert ok, err


@pytest.mark.parametrize("case", [(1000, 1, 1), (777, 3, 4), (513, 12, 16), (300, 64, 64), (129, 200, 200), (70, 1024, 1024), (5, 96, 100), (40000, 32, 32)])
def test_bias_grad_partial_column_sums(backend, case):
    """mh_bias_grad_partial (ABI 15): the same column sums as per-workgroup partial rows ws[nblocks][nch] + one mh_wgrad_reduce segment that sums them in order
    onto db (accumulating) -- no float atomics: two runs are bit-identical, the sum is the oracle's."""
    npix, nch, ld = case
    dev = backend.device
    t = _rand((1, 1, npix, ld), 31, dev)
    v = ops.View(t, 1, 1, npix, nch, ld)
    db0 = _rand((nch,), 32, dev)
    ref = db0.cpu().double() + t.cpu().double()[0, 0, :, :nch].sum(0)
    outs = []
    for _ in range(2):
        db = db0.clone()
        wsa = ops.WgradWorkspace(dev); segs, keep = [], []
        ops.bias_grad_partial(backend.lib, backend.lib, wsa, segs, v, db)
        assert len(segs) == 1 and segs[0][2] == nch and segs[0][3] == backend.lib.bias_grad_blocks(npix, nch) and segs[0][4] == 1
        ops.wgrad_reduce(backend.lib, segs, dev, keep)
        backend.sync()
        outs.append(db.cpu())
    assert torch.equal(outs[0], outs[1])
    assert (outs[0].double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())


def test_deterministic_twin_reports_saturated_addends(backend):
    """mh_deterministic_overflow: an addend beyond the fixed-point twin's range (|v| >= 2^15) is saturated AND flagged; in-range sums are not."""
    import ctypes as C
    dev = backend.device
    db = torch.zeros(4, device=dev); twin = torch.zeros(4, dtype=torch.int64, device=dev)
    assert backend.lib.deterministic_overflow() == 0
    assert backend.lib.deterministic_add(C.c_void_p(db.data_ptr()), 4, C.c_void_p(twin.data_ptr())) == 0
    try:
        small = torch.full((1, 1, 64, 4), 2.0, device=dev)
        ops.bias_grad(backend.lib, ops.View(small, 1, 1, 64, 4, 4), db)
        backend.sync()
        assert backend.lib.deterministic_overflow() == 0
        big = torch.full((1, 1, 64, 4), 1.0e5, device=dev)                 # a workgroup's column sum = 6.4e6 >> 32767
        ops.bias_grad(backend.lib, ops.View(big, 1, 1, 64, 4, 4), db)
        backend.sync()
        assert backend.lib.deterministic_overflow() == 1 and backend.lib.deterministic_overflow() == 0        # sticky until read, then cleared
    finally:
        backend.lib.deterministic_remove(C.c_void_p(db.data_ptr()))


@pytest.mark.parametrize("case", [(1000, 1, 1), (777, 3, 4), (513, 12, 16), (300, 64, 64), (129, 200, 200), (70, 1024, 1024), (5, 96, 100)])
def test_bias_grad_column_sums(backend, case):
    """db += column sums of a [npix, nch] view with row stride ld (BiasAddGrad of the transposed convs)."""
    npix, nch, ld = case
    dev = backend.device
    t = _rand((1, 1, npix, ld), 31, dev)
    v = ops.View(t, 1, 1, npix, nch, ld)
    db = _rand((nch,), 32, dev)
    ref = db.cpu().double() + t.cpu().double()[0, 0, :, :nch].sum(0)
    ops.bias_grad(backend.lib, v, db)
    backend.sync()
    assert (db.cpu().double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("shape", [(1, 9, 14), (2, 12, 21), (1, 40, 70)])
def test_proxy_loss_and_grad(backend, shape):
    """Continual-adaptation loss: weight * masked mean L1 against proxy labels; proxies <= 0 or >= 192 are invalid."""
    B, H, W = shape
    dev = backend.device
    g = torch.Generator().manual_seed(17)
    pred = (torch.rand(B, H, W, generator=g) * 200 - 4).to(dev)
    proxy = (torch.rand(B, H, W, generator=g) * 230 - 20)
    proxy[0, 0, :3] = torch.tensor([0.0, 192.0, 191.99])          # boundary values of the validity test
    pred[0, 1, 0] = proxy[0, 1, 0]                                  # |x| at 0: zero gradient
    proxy = proxy.to(dev)
    pc = pred.cpu().requires_grad_(True)
    ref = T.proxy_loss(pc, proxy.cpu(), 0.1)
    (gref,) = torch.autograd.grad(ref, [pc])
    ws = torch.zeros(backend.lib.proxy_ws_floats(B, H, W), device=dev)
    res = torch.zeros(4, device=dev); dp = torch.full((B, H, W), float("nan"), device=dev)
    ops.proxy_loss(backend.lib, pred, proxy, ws, res, dp, weight=0.1, grad_scale=1.0)
    backend.sync()
    assert abs(res[0].item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item()))
    valid = ((proxy.cpu() > 0) & (proxy.cpu() < 192)).sum().item()
    assert res[1].item() == float(valid)
    ok, err = _close(dp, gref, rtol=1e-5, atol=1e-9); assert ok, err


@pytest.mark.parametrize("shape", [(1, 9, 14), (2, 11, 23)])
def test_supervised_loss_and_grad(backend, shape):
    """Train.py's loss term of one scale: weight * masked mean L1 against the ground truth; gt == 0 or >= max_disp is invalid
    (a NEGATIVE gt stays valid -- unlike the proxy rule)."""
    B, H, W = shape
    dev = backend.device
    g = torch.Generator().manual_seed(23)
    pred = (torch.rand(B, H, W, generator=g) * 200 - 4).to(dev)
    gt = torch.rand(B, H, W, generator=g) * 230 - 20
    gt[0, 0, :4] = torch.tensor([0.0, 192.0, 191.99, -0.0])
    pred[0, 1, 0] = gt[0, 1, 0]
    gt = gt.to(dev)
    pc = pred.cpu().requires_grad_(True)
    ref = T.supervised_loss(pc, gt.cpu(), 0.7, 192.0)
    (gref,) = torch.autograd.grad(ref, [pc])
    ws = torch.zeros(backend.lib.proxy_ws_floats(B, H, W), device=dev)
    res = torch.zeros(4, device=dev); dp = torch.full((B, H, W), float("nan"), device=dev)
    ops.supervised_loss(backend.lib, pred, gt, ws, res, dp, weight=0.7, grad_scale=1.0, max_disp=192.0)
    backend.sync()
    assert abs(res[0].item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item()))
    assert res[1].item() == float(((gt.cpu() != 0) & (gt.cpu() < 192)).sum().item())
    ok, err = _close(dp, gref, rtol=1e-5, atol=1e-9); assert ok, err


def test_adam_update(backend):
    """tf.train.AdamOptimizer(lr, 0.9): three steps over two variable ranges sharing one beta-power state, through the plan
    recorder (the way the engine issues them) and eagerly -- against the oracle's fp32 restatement."""
    from madnet_hip import plan as PL
    dev = backend.device
    n = 777
    w = _rand((n,), 51, dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    state = torch.tensor([0.9, 0.999], device=dev)
    wr, mr, vr, sr = w.cpu().clone(), torch.zeros(n), torch.zeros(n), [0.9, 0.999]
    for t in range(3):
        g = _rand((n,), 60 + t, dev, 0.3)
        if t == 1:                                   # recorded + replayed
            r = PL.Recorder()
            ops.adam(r, w[:300], m[:300], v[:300], g[:300], state, lr=1e-3, n=300)
            ops.adam(r, w[300:], m[300:], v[300:], g[300:], state, lr=1e-3, n=n - 300)
            ops.adam_advance(r, state)
            r.keep.append(g)
            r.compile().run(backend.lib, 0)
        else:
            ops.adam(backend.lib, w, m, v, g, state, lr=1e-3)
            ops.adam_advance(backend.lib, state)
        backend.sync()
        T.adam_update(wr, mr, vr, g.cpu(), sr, 1e-3)
        sr[0] = float(torch.tensor(sr[0], dtype=torch.float32) * torch.tensor(0.9, dtype=torch.float32))
        sr[1] = float(torch.tensor(sr[1], dtype=torch.float32) * torch.tensor(0.999, dtype=torch.float32))
        assert torch.allclose(w.cpu(), wr, rtol=2e-6, atol=1e-7), (t, (w.cpu() - wr).abs().max())
        assert torch.allclose(v.cpu(), vr, rtol=1e-6, atol=1e-12) and torch.allclose(m.cpu(), mr, rtol=1e-6, atol=1e-9)
        assert torch.allclose(state.cpu(), torch.tensor(sr), rtol=1e-7)



@pytest.mark.parametrize("case", [(1, 12, 40, 128, 2), (2, 6, 10, 32, 2), (1, 9, 17, 64, 3), (1, 24, 80, 96, 2), (1, 5, 7, 16, 1)])
def test_level_front_fused_matches_resize_warp_corr(backend, case):
    """mh_level_front_fwd (one launch) == mh_resize_fwd(mode 0) -> mh_warp_fwd -> mh_corr_fwd(copy_left, u, zero_tail), the
    three launches it replaces, and == the oracle chain (tf.image.resize_images * 20/2^k, _linear_warping, correlation + concat;
    MadNet.py:274-295,370-436).  Large upsampled disparities push taps outside the row (zero weight)."""
    B, H, W, C, md = case
    dev = backend.device
    Hc, Wc = (H + 1) // 2, (W + 1) // 2
    Vc = _rand((B, Hc, Wc), 61, dev, 1.5)
    L = _rand((B, H, W, C), 62, dev); R = _rand((B, H, W, C), 63, dev)
    mul = 20.0 / 8
    D = 2 * md + 1
    ld = (C + D + 1 + 3) // 4 * 4
    # unfused chain
    u0 = torch.empty(B, H, W, device=dev); Rw0 = torch.empty(B, H, W, C, device=dev)
    out0 = torch.full((B, H, W, ld), float("nan"), device=dev)
    ops.resize_fwd(backend.lib, Vc, u0, H, W, mul=mul, mode=0)
    ops.warp_fwd(backend.lib, ops.view(R), u0, ops.view(Rw0))
    ops.corr_fwd(backend.lib, ops.view(L), ops.view(Rw0), ops.View(out0, B, H, W, ld, ld), md, 1, coff=C, u=u0, copy_left=True, zero_tail=True)
    # fused
    u1 = torch.full((B, H, W), float("nan"), device=dev); Rw1 = torch.full((B, H, W, C), float("nan"), device=dev)
    out1 = torch.full((B, H, W, ld), float("nan"), device=dev)
    ops.level_front_fwd(backend.lib, Vc, mul, ops.view(L), ops.view(R), ops.View(out1, B, H, W, ld, ld), ops.view(Rw1), u1, md, coff=C)
    backend.sync()
    # u: both kernels run the interpolation un-contracted -> a few ulps at most; the warped features then differ by
    # |du| * |dR/dx| (the warp is continuous in u, slope |R[x0+1] - R[x0]| ~ 3 here) plus fma-contraction noise
    du = (u1.cpu() - u0.cpu()).abs().max().item()
    assert du <= 1e-5
    assert (Rw1.cpu() - Rw0.cpu()).abs().max().item() <= 1e-5 + 8.0 * du
    assert torch.equal(out1[..., :C].cpu(), L.cpu()) and torch.equal(out1[..., C + D].cpu(), u1.cpu())
    assert torch.all(out1[..., C + D + 1:].cpu() == 0)
    assert (out1[..., C:C + D].cpu() - out0[..., C:C + D].cpu()).abs().max().item() <= 1e-5 + 8.0 * du
    # oracle
    uo = T.resize_bilinear(Vc.cpu()[..., None], H, W) * mul
    ref = T.correlation(L.cpu(), T.linear_warp(R.cpu(), uo), md, 1)
    ok, err = _close(out1[..., C:C + D], ref); assert ok, err


@pytest.mark.parametrize("case", [(1, 12, 40, 128, 2, 32), (2, 6, 10, 32, 2, 32), (1, 9, 17, 64, 3, 32), (1, 24, 80, 96, 2, 32), (1, 5, 7, 16, 1, 16), (1, 48, 160, 32, 2, 32),
                                  (1, 2, 4, 128, 2, 32)])
def test_level_front_with_the_coarser_head_inside(backend, case):
    """mh_level_front_head_fwd (the disparity head of the coarser level computed IN the front-end launch, MadNet.py:118 then :274-295) == the head as a launch
    of its own (mh_conv2d_fwd, 3x3 K -> 1, linear) followed by mh_level_front_fwd: the stored Vc bit for bit where the two kernels contract alike (asserted to
    2 ulp-of-sum), u / warped features / cost volume to the interpolation's noise; every element of Vc is written (NaN canary); odd sizes, two images, a row
    shorter than a workgroup; and the planes of the estimator input.  Against the oracle: conv2d + resize_images + _linear_warping + correlation."""
    B, H, W, C, md, K = case
    dev = backend.device
    lib = backend.lib
    Hc, Wc = (H + 1) // 2, (W + 1) // 2
    assert lib.level_front_head_ok(Hc, Wc, H, W, C, K, md) == 1
    X = _rand((B, Hc, Wc, K), 71, dev)
    hw = _rand((3, 3, K, 1), 72, dev, 0.2)
    hb = _rand((1,), 73, dev)
    L = _rand((B, H, W, C), 62, dev); R = _rand((B, H, W, C), 63, dev)
    mul = 20.0 / 8
    D = 2 * md + 1
    ld = (C + D + 1 + 3) // 4 * 4
    # the two launches
    V0 = torch.full((B, Hc, Wc), float("nan"), device=dev)
    ops.conv2d_fwd(lib, ops.view(X), hw, hb, ops.View(V0, B, Hc, Wc, 1, 1), alpha=1.0)
    assert "conv_n1_fwd_kernel" in lib.last_kernel().decode()
    u0 = torch.full((B, H, W), float("nan"), device=dev); Rw0 = torch.full((B, H, W, C), float("nan"), device=dev)
    out0 = torch.full((B, H, W, ld), float("nan"), device=dev)
    ops.level_front_fwd(lib, V0, mul, ops.view(L), ops.view(R), ops.View(out0, B, H, W, ld, ld), ops.view(Rw0), u0, md, coff=C)
    # one launch
    V1 = torch.full((B, Hc, Wc), float("nan"), device=dev)
    u1 = torch.full((B, H, W), float("nan"), device=dev); Rw1 = torch.full((B, H, W, C), float("nan"), device=dev)
    out1 = torch.full((B, H, W, ld), float("nan"), device=dev)
    pl = ops.Planes(ops.Shadow(B, H, W, C + D + 1, dev), dev)
    ops.level_front_head_fwd(lib, ops.view(X), hw, hb, V1, mul, ops.view(L), ops.view(R), ops.View(out1, B, H, W, ld, ld), ops.view(Rw1), u1, md, coff=C, planes=pl)
    assert "HEAD=%d" % K in lib.last_kernel().decode()
    backend.sync()
    v0, v1 = V0.cpu(), V1.cpu()
    assert torch.isfinite(v1).all()                                       # every coarse pixel was stored by some workgroup
    dv = (v1 - v0).abs().max().item()
    assert dv <= 2e-6 * max(1.0, v0.abs().max().item()), dv
    du = (u1.cpu() - u0.cpu()).abs().max().item()
    assert du <= 1e-5 + 4.0 * mul * dv
    assert (Rw1.cpu() - Rw0.cpu()).abs().max().item() <= 1e-5 + 32.0 * du       # |du| x the steepest |R[x+1] - R[x]| of the map
    assert torch.equal(out1[..., :C].cpu(), L.cpu()) and torch.equal(out1[..., C + D].cpu(), u1.cpu())
    assert torch.all(out1[..., C + D + 1:].cpu() == 0)
    assert (out1[..., C:C + D].cpu() - out0[..., C:C + D].cpu()).abs().max().item() <= 1e-5 + 32.0 * du
    if H == 2 * Hc:
        # the default instance gave two fine rows to a workgroup; one row per workgroup (mh_tune_corr bit 3) must store the same bits
        lib.tune_corr(1 | 8)
        try:
            V2 = torch.full((B, Hc, Wc), float("nan"), device=dev); u2 = torch.empty(B, H, W, device=dev); Rw2 = torch.empty(B, H, W, C, device=dev)
            out2 = torch.full((B, H, W, ld), float("nan"), device=dev)
            ops.level_front_head_fwd(lib, ops.view(X), hw, hb, V2, mul, ops.view(L), ops.view(R), ops.View(out2, B, H, W, ld, ld), ops.view(Rw2), u2, md, coff=C)
            backend.sync()
        finally:
            lib.tune_corr(1)
        assert torch.equal(V2.cpu(), v1) and torch.equal(u2.cpu(), u1.cpu()) and torch.equal(out2.cpu(), out1.cpu()) and torch.equal(Rw2.cpu(), Rw1.cpu())
    # the planes: hi + lo of what the launch stored in fp32
    hi = pl.hi.t.float().cpu()[..., :C + D + 1]
    lo = pl.lo.t.float().cpu()[..., :C + D + 1]
    assert ((hi + lo) - out1[..., :C + D + 1].cpu()).abs().max().item() <= 2e-5 * max(1.0, out1[..., :C + D + 1].abs().max().item())
    # oracle
    Vo = T.conv2d(X.cpu(), hw.cpu(), hb.cpu(), 1, 1, 1.0)
    ok, err = _close(V1, Vo[..., 0]); assert ok, err
    uo = T.resize_bilinear(Vo, H, W) * mul
    ref = T.correlation(L.cpu(), T.linear_warp(R.cpu(), uo), md, 1)
    ok, err = _close(out1[..., C:C + D], ref); assert ok, err


def test_level_front_head_refuses_what_it_does_not_serve(backend):
    """K > 32 head channels / D > 9: mh_level_front_head_ok says 0 and the entry point answers MH_ERR_UNSUPPORTED instead of running something else"""
    lib = backend.lib
    assert lib.level_front_head_ok(6, 10, 12, 20, 32, 64, 2) == 0 and lib.level_front_head_ok(6, 10, 12, 20, 32, 30, 2) == 0
    assert lib.level_front_head_ok(6, 10, 12, 20, 32, 32, 5) == 0 and lib.level_front_head_ok(6, 10, 12, 20, 32, 32, 2) == 1
    # a down-scaling geometry would leave coarse pixels no fine pixel interpolates from un-stored: not served
    assert lib.level_front_head_ok(12, 20, 6, 10, 32, 32, 2) == 0 and lib.level_front_head_ok(6, 20, 12, 10, 32, 32, 2) == 0
    dev = backend.device
    B, H, W, C, md, K = 1, 12, 20, 32, 2, 64
    X = _rand((B, 6, 10, K), 1, dev); hw = _rand((3, 3, K, 1), 2, dev); V = torch.zeros(B, 6, 10, device=dev)
    L = _rand((B, H, W, C), 3, dev); R = _rand((B, H, W, C), 4, dev)
    ld = 40
    out = torch.zeros(B, H, W, ld, device=dev); Rw = torch.zeros(B, H, W, C, device=dev); u = torch.zeros(B, H, W, device=dev)
    with pytest.raises(Exception) as e:
        ops.level_front_head_fwd(lib, ops.view(X), hw, None, V, 2.5, ops.view(L), ops.view(R), ops.View(out, B, H, W, ld, ld), ops.view(Rw), u, md, coff=C)
    assert "mh_level_front_head_ok" in str(e.value)


@pytest.mark.parametrize("case", [(1, 2, 70, 128, 40), (2, 1, 131, 64, 40), (1, 2, 40, 32, 10), (1, 1, 200, 256, 24)])
def test_corr_fwd_large_d_bf16_and_split_bf16(backend, case):
    """DispNet's 81-shift volume on the bf16 matrix cores (mh_corr_fwd_prec): precision 1 = operands rounded to bf16 (judged
    against the oracle on bf16-rounded inputs: products of bf16 values are exact in fp32), precision 2 = split-bf16 (judged against
    the UNROUNDED oracle: ~2^-16 relative); ragged row ends, both borders."""
    B, H, W, C, md = case
    dev = backend.device
    L = _rand((B, H, W, C), 71, dev); R = _rand((B, H, W, C), 72, dev)
    D = 2 * md + 1
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
    ref = T.correlation(L.cpu().double(), R.cpu().double(), md, 1).float()
    ref_bf = T.correlation(bf(L.cpu()), bf(R.cpu()), md, 1)
    out1 = torch.full((B, H, W, D), float("nan"), device=dev); out2 = torch.full((B, H, W, D), float("nan"), device=dev)
    ops.corr_fwd(backend.lib, ops.view(L), ops.view(R), ops.view(out1), md, 1, precision=1)
    k1 = backend.lib.last_kernel().decode()
    ops.corr_fwd(backend.lib, ops.view(L), ops.view(R), ops.view(out2), md, 1, precision=2)
    k2 = backend.lib.last_kernel().decode()
    backend.sync()
    assert "bf16" in k1 and "bf16x3" in k2, (k1, k2)
    assert (out1.cpu() - ref_bf).abs().max().item() <= 2e-6
    e2 = (out2.cpu() - ref).abs().max().item(); e1 = (out1.cpu() - ref).abs().max().item()
    assert e2 <= 5e-6 and e2 * 30 <= e1, (e1, e2)          # |corr| ~ 0.3: 2^-16 relative per product, averaged over C channels


@pytest.mark.parametrize("form", ["rowlds", "row", "atomic"])
@pytest.mark.parametrize("case", [(1, 12, 40, 32, 2, 1), (2, 9, 21, 96, 2, 1), (1, 7, 33, 64, 2, 1), (1, 6, 20, 192, 2, 1), (1, 10, 17, 16, 1, 1), (1, 3, 300, 32, 2, 1)])
def test_corr_warp_bwd_fused_matches_corr_bwd_then_warp_bwd(backend, case, form):
    """mh_corr_warp_bwd = mh_corr_bwd (warped right features as the right operand, fused concat form) + mh_warp_bwd of its result in one launch: dL and the
    coordinate gradient du are compared with the two-launch sequence at rounding level, the scatter with the tolerance of its summation order.
    form: 'rowlds' = the row-owned kernel with the row's operands staged in LDS (round 5: the default), 'row' = row-owned, operands from L1 / L2
    (mh_tune_corr_row(3); also what 'rowlds' falls back to beyond 128 channels), 'atomic' = the global-atomic kernel (mh_tune_corr_row(0))."""
    B, H, W, Cc, md, stride = case
    dev = backend.device
    D = 2 * md // stride + 1
    ld = (Cc + D + 1 + 3) // 4 * 4
    g = torch.randn(B, H, W, ld, device=dev)
    L = torch.randn(B, H, W, Cc, device=dev); R = torch.randn(B, H, W, Cc, device=dev)
    u = (torch.rand(B, H, W, device=dev) - 0.5) * 6.0
    if W >= 100:
        u = u * 20.0                      # taps far from their pixel, many of them clamped / masked at the row ends
    Rw = torch.zeros(B, H, W, Cc, device=dev)
    ops.warp_fwd(backend.lib, ops.view(R), u, ops.view(Rw))
    gv = ops.View(g, B, H, W, ld, ld)
    outs = []
    backend.lib.tune_corr_row({"rowlds": 1, "row": 3, "atomic": 0}[form])
    expect = {"rowlds": "corr_warp_bwd_rowlds_kernel" if Cc <= 128 else "corr_warp_bwd_row_kernel", "row": "corr_warp_bwd_row_kernel", "atomic": "corr_warp_bwd_kernel"}[form]
    try:
        for fused in (False, True):
            dL = torch.full((B, H, W, Cc), 0.25, device=dev)                 # acc_l: accumulate onto an earlier contribution
            dimg = torch.full((B, H, W, Cc), -0.5, device=dev)              # the scatter target holds an earlier contribution too
            du = torch.full((B, H, W), float("nan"), device=dev)
            if fused:
                ops.corr_warp_bwd(backend.lib, gv, ops.view(L), ops.view(Rw), ops.view(R), u, ops.view(dL), ops.view(dimg), du, md, stride, coff=Cc, acc_l=True, copy_left=True)
                assert expect in backend.lib.last_kernel().decode()
            else:
                dRw = torch.zeros(B, H, W, Cc, device=dev)
                ops.corr_bwd(backend.lib, gv, ops.view(L), ops.view(Rw), ops.view(dL), ops.view(dRw), md, stride, coff=Cc, du=du, acc_l=True, acc_r=False, acc_u=False, copy_left=True)
                ops.warp_bwd(backend.lib, ops.view(dRw), ops.view(R), u, ops.view(dimg), du=du, acc_u=True)
            backend.sync()
            outs.append((dL.cpu(), dimg.cpu(), du.cpu()))
        (dL0, di0, du0), (dL1, di1, du1) = outs
        assert torch.isfinite(du1).all()
        assert (dL0 - dL1).abs().max().item() <= 1e-6 * max(1.0, dL0.abs().max().item())
        assert (du0 - du1).abs().max().item() <= 2e-5 * max(1.0, du0.abs().max().item())
        assert (di0 - di1).abs().max().item() <= 2e-5 * max(1.0, di0.abs().max().item())
        # du only / scatter only
        du2 = torch.full((B, H, W), float("nan"), device=dev); dL2 = torch.zeros(B, H, W, Cc, device=dev)
        ops.corr_warp_bwd(backend.lib, gv, ops.view(L), ops.view(Rw), ops.view(R), u, ops.view(dL2), None, du2, md, stride, coff=Cc, copy_left=True)
        dimg3 = torch.full((B, H, W, Cc), -0.5, device=dev); dL3 = torch.zeros(B, H, W, Cc, device=dev)
        ops.corr_warp_bwd(backend.lib, gv, ops.view(L), ops.view(Rw), ops.view(R), u, ops.view(dL3), ops.view(dimg3), None, md, stride, coff=Cc, copy_left=True)
        backend.sync()
        assert (du2.cpu() - du1).abs().max().item() <= 2e-5 * max(1.0, du1.abs().max().item())
        assert (dimg3.cpu() - di1).abs().max().item() <= 2e-5 * max(1.0, di1.abs().max().item())
        assert (dL3.cpu() - dL2.cpu()).abs().max().item() == 0.0
    finally:
        backend.lib.tune_corr_row(1)


def test_corr_warp_bwd_many_taps_on_one_column(backend):
    """A warp that sends MANY pixels to the same source column (u = const - x over a span: 12 / 40 taps per column, more than the 8 slots of a
    column's tap list in the row kernel): the gather falls back to scanning the row for such columns; result = the two-launch sequence."""
    B, H, W, Cc, md = 1, 3, 40, 32, 2
    dev = backend.device
    D = 2 * md + 1
    ld = (Cc + D + 1 + 3) // 4 * 4
    g = torch.randn(B, H, W, ld, device=dev)
    L = torch.randn(B, H, W, Cc, device=dev); R = torch.randn(B, H, W, Cc, device=dev)
    xs = torch.arange(W, dtype=torch.float32)
    u = torch.zeros(B, H, W)
    u[0, 0] = 17.3 - xs                                  # the whole row lands on columns 17 / 18
    u[0, 1, 10:22] = 5.5 - xs[10:22]                     # twelve pixels on columns 5 / 6, the rest in place
    u[0, 2] = (torch.rand(W) - 0.5) * 3.0
    u = u.to(dev)
    Rw = torch.zeros(B, H, W, Cc, device=dev)
    ops.warp_fwd(backend.lib, ops.view(R), u, ops.view(Rw))
    gv = ops.View(g, B, H, W, ld, ld)
    dL0 = torch.zeros(B, H, W, Cc, device=dev); dRw = torch.zeros(B, H, W, Cc, device=dev); di0 = torch.zeros(B, H, W, Cc, device=dev); du0 = torch.zeros(B, H, W, device=dev)
    ops.corr_bwd(backend.lib, gv, ops.view(L), ops.view(Rw), ops.view(dL0), ops.view(dRw), md, 1, coff=Cc, du=du0, copy_left=True)
    ops.warp_bwd(backend.lib, ops.view(dRw), ops.view(R), u, ops.view(di0), du=du0, acc_u=True)
    dL1 = torch.zeros(B, H, W, Cc, device=dev); di1 = torch.zeros(B, H, W, Cc, device=dev); du1 = torch.zeros(B, H, W, device=dev)
    ops.corr_warp_bwd(backend.lib, gv, ops.view(L), ops.view(Rw), ops.view(R), u, ops.view(dL1), ops.view(di1), du1, md, 1, coff=Cc, copy_left=True)
    backend.sync()
    assert "corr_warp_bwd_rowlds_kernel" in backend.lib.last_kernel().decode()
    assert (di0.cpu() - di1.cpu()).abs().max().item() <= 2e-5 * max(1.0, di0.abs().max().item())
    assert (du0.cpu() - du1.cpu()).abs().max().item() <= 2e-5 * max(1.0, du0.abs().max().item())
    assert int((di0[0, 0].abs().sum(-1) > 0).sum().item()) == 2 and int((di0[0, 2].abs().sum(-1) > 0).sum().item()) > 30       # (row 0: every tap on two columns)
    # a fold of 40 taps onto one column is summed in ascending (pixel, tap) order like any other (beyond 16 taps it used to be arrival order):
    # a second launch gives the same bits
    di2 = torch.zeros(B, H, W, Cc, device=dev); dL2 = torch.zeros(B, H, W, Cc, device=dev); du2 = torch.zeros(B, H, W, device=dev)
    ops.corr_warp_bwd(backend.lib, gv, ops.view(L), ops.view(Rw), ops.view(R), u, ops.view(dL2), ops.view(di2), du2, md, 1, coff=Cc, copy_left=True)
    backend.sync()
    assert torch.equal(di1.cpu(), di2.cpu()) and torch.equal(du1.cpu(), du2.cpu())


def test_corr_warp_bwd_row_form_is_deterministic_without_a_twin(backend):
    """The default row-owned kernel GATHERS the warp-gradient taps in a fixed order: two launches on the same operands give bit-identical results, with
    or without a registered deterministic range (the global-atomic form needs the range's fixed-point twin + mh_det_flush for that).  The scatter form
    of the row kernel (mh_tune_corr_row(3)) switches to 64-bit fixed-point LDS accumulators when a range is registered: also bit-identical."""
    B, H, W, Cc, md = 1, 6, 70, 32, 2
    dev = backend.device
    D = 2 * md + 1
    ld = (Cc + D + 1 + 3) // 4 * 4
    g = torch.randn(B, H, W, ld, device=dev)
    L = torch.randn(B, H, W, Cc, device=dev); R = torch.randn(B, H, W, Cc, device=dev)
    u = (torch.rand(B, H, W, device=dev) - 0.5) * 9.0
    Rw = torch.zeros(B, H, W, Cc, device=dev)
    ops.warp_fwd(backend.lib, ops.view(R), u, ops.view(Rw))
    gv = ops.View(g, B, H, W, ld, ld)

    def run():
        dL = torch.zeros(B, H, W, Cc, device=dev); dimg = torch.full((B, H, W, Cc), 0.125, device=dev); du = torch.zeros(B, H, W, device=dev)
        ops.corr_warp_bwd(backend.lib, gv, ops.view(L), ops.view(Rw), ops.view(R), u, ops.view(dL), ops.view(dimg), du, md, 1, coff=Cc, copy_left=True)
        backend.sync()
        return dimg.cpu(), backend.lib.last_kernel().decode()

    plain, k0 = run()
    again, _ = run()
    assert "corr_warp_bwd_rowlds_kernel" in k0 and torch.equal(plain, again)
    import ctypes as C
    other = torch.zeros(64, device=dev); twin = torch.zeros(64, dtype=torch.int64, device=dev)          # an unrelated registered range switches the mode on
    assert backend.lib.deterministic_add(C.c_void_p(other.data_ptr()), 64, C.c_void_p(twin.data_ptr())) == 0
    try:
        a, k1 = run()
        backend.lib.tune_corr_row(3)
        b, k2 = run()
        c, _ = run()
    finally:
        backend.lib.tune_corr_row(1)
        backend.lib.deterministic_remove(C.c_void_p(other.data_ptr()))
    assert "corr_warp_bwd_rowlds_kernel" in k1 and torch.equal(a, plain)
    assert "corr_warp_bwd_row_kernel" in k2 and ",det" in k2 and torch.equal(b, c)
    assert (b - plain).abs().max().item() <= 2e-5 * max(1.0, plain.abs().max().item())


@pytest.mark.parametrize("case", [(1, 2, 70, 128, 40), (2, 1, 131, 64, 40), (1, 2, 40, 32, 10), (1, 1, 200, 256, 24)])
def test_corr_bwd_large_d_bf16(backend, case):
    """Gradient of DispNet's 81-shift volume on the bf16 matrix cores (mh_corr_bwd_prec, precision 1): judged against autograd of the TF formulation
    (sharedLayers.py:41-51) on bf16-ROUNDED g, L, R (products of bf16 values are exact in fp32: only the summation order is left), and against the exact
    fp32 kernel pair at bf16 level; ragged row ends, both borders, accumulate flags, a g buffer with channels behind the volume."""
    B, H, W, C, md = case
    dev = backend.device
    D = 2 * md + 1
    ld = (D + 5 + 3) // 4 * 4
    L = _rand((B, H, W, C), 81, dev); R = _rand((B, H, W, C), 82, dev)
    g = _rand((B, H, W, ld), 83, dev)
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
    Lr = bf(L.cpu()).double().requires_grad_(True); Rr = bf(R.cpu()).double().requires_grad_(True)
    out = T.correlation(Lr, Rr, md, 1)
    (out * bf(g.cpu()[..., :D]).double()).sum().backward()
    gv = ops.View(g, B, H, W, D, ld)
    dL = torch.full((B, H, W, C), 0.5, device=dev); dR = torch.full((B, H, W, C), float("nan"), device=dev)
    ops.corr_bwd(backend.lib, gv, ops.view(L), ops.view(R), ops.view(dL), ops.view(dR), md, 1, coff=0, acc_l=True, acc_r=False, precision=1)
    k = backend.lib.last_kernel().decode()
    backend.sync()
    assert "corr_bwd_mfma_bf16_pair" in k, k
    sc = max(1.0, Lr.grad.abs().max().item())
    assert (dL.cpu() - 0.5 - Lr.grad.float()).abs().max().item() <= 2e-6 * sc
    assert (dR.cpu() - Rr.grad.float()).abs().max().item() <= 2e-6 * sc
    dL0 = torch.zeros(B, H, W, C, device=dev); dR0 = torch.zeros(B, H, W, C, device=dev)
    ops.corr_bwd(backend.lib, gv, ops.view(L), ops.view(R), ops.view(dL0), ops.view(dR0), md, 1, coff=0, precision=0)
    backend.sync()
    assert "bf16" not in backend.lib.last_kernel().decode()
    assert (dL.cpu() - 0.5 - dL0.cpu()).abs().max().item() <= 2e-2 * sc and (dR.cpu() - dR0.cpu()).abs().max().item() <= 2e-2 * sc


@pytest.mark.parametrize("case", [(1, 12, 40, 32), (2, 9, 21, 32), (1, 3, 5, 32), (1, 24, 80, 16), (1, 1, 2, 32)])
@pytest.mark.parametrize("kind", ["du", "addends"])
def test_head_bwd_matches_the_separate_launches(backend, case, kind):
    """mh_head_bwd = mh_resize_bwd (mode 0, the x2 legacy resize of the disparity) or the per-pixel sum of two strided addends, followed by the
    input gradient of the 3x3 Cin -> 1 head conv with the leaky mask of the layer below -- one launch, dV and dx bit-identical to the sequence
    (same arithmetic, same order), bf16 shadows of both written on the way."""
    B, H, W, N = case
    dev = backend.device
    w = torch.randn(3, 3, N, 1, device=dev) * 0.3
    mref = torch.randn(B, H, W, N + 4, device=dev)
    mv = ops.View(mref, B, H, W, N, N + 4)
    old = torch.randn(B, H, W, N, device=dev)
    V = torch.randn(B, H, W, device=dev)
    outs = []
    for fused in (False, True):
        dV = torch.full((B, H, W), float("nan"), device=dev)
        dxb = torch.full((B, H, W, N + 4), 7.0, device=dev)
        dxb[..., :N] = old
        dxv = ops.View(dxb, B, H, W, N, N + 4)
        shV, shx = ops.Shadow(B, H, W, 1, dev), ops.Shadow(B, H, W, N, dev)
        if kind == "du":
            du = torch.randn(B, 2 * H, 2 * W, device=dev)
            torch.manual_seed(5); du = torch.randn(B, 2 * H, 2 * W, device=dev)
            if fused:
                ops.head_bwd(backend.lib, w, dV, dxv, mask_ref=mv, mask_alpha=0.2, accumulate_dx=True, du=du, Hr=2 * H, Wr=2 * W, mul=2.5, dV_shadow=shV, dx_shadow=shx)
            else:
                ops.resize_bwd(backend.lib, du, V, dV, 2 * H, 2 * W, mul=2.5, mode=0, accumulate=False)
        else:
            torch.manual_seed(6)
            a1 = torch.randn(B, H, W, device=dev); a2b = torch.randn(B, H, W, 8, device=dev)
            a2 = ops.View(a2b, B, H, W, 8, 8).slice(5, 6)
            if fused:
                ops.head_bwd(backend.lib, w, dV, dxv, mask_ref=mv, mask_alpha=0.2, accumulate_dx=True, addends=(ops.view(a1), a2), dV_shadow=shV, dx_shadow=shx)
            else:
                ops.copy_channels(backend.lib, ops.view(a1), ops.view(dV))
                ops.copy_channels(backend.lib, a2, ops.view(dV), accumulate=True)
        if not fused:
            ops.conv2d_dgrad(backend.lib, ops.view(dV), w, dxv, accumulate=True, mask_ref=mv, mask_alpha=0.2, shadow=shx)
        else:
            assert "head_bwd_kernel" in backend.lib.last_kernel().decode()
        backend.sync()
        assert (dxb[..., N:] == 7.0).all()
        outs.append((dV.cpu().clone(), dxb[..., :N].cpu().clone(), shV.t.cpu().clone(), shx.t.cpu().clone()))
    (dV0, dx0, _, shx0), (dV1, dx1, shV1, shx1) = outs
    assert torch.equal(dV0, dV1)
    assert (dx0 - dx1).abs().max().item() <= 1e-6 * max(1.0, dx0.abs().max().item())
    assert torch.equal(shV1[..., 0], dV1.to(torch.bfloat16)) and (shV1[..., 1:] == 0).all()
    assert torch.equal(shx1[..., :N], dx1.to(torch.bfloat16))


def test_fetch_inputs_table_direct_and_as_plan_op(backend):
    """mh_fetch_inputs / MH_OP_FETCH_INPUTS: the step's first node fills its fixed input buffers from the tensors a host-rewritten table names -- uint8 sources are cast
    (tf.cast of Data_utils/data_reader.py:98), float32 copied, an empty entry or the buffer itself left alone; sizes that are not multiples of four; the SAME recorded
    plan follows the table from run to run (what a captured step does with a prefetcher's rotating slots)."""
    from madnet_hip.plan import Recorder
    lib, dev = backend.lib, backend.device
    g = torch.Generator().manual_seed(5)
    n_img, n_gt = 3 * 7 * 11 * 3 + 2, 7 * 11 + 1          # not multiples of 4
    dst = [torch.full((n,), -7.0, device=dev) for n in (n_img, n_img, n_gt, n_gt)]
    keep3 = dst[3].clone()
    tab = ops.InputTable(lib, dev)

    def frames(seed):
        gg = torch.Generator().manual_seed(seed)
        return (torch.randint(0, 256, (n_img,), generator=gg, dtype=torch.uint8).to(dev), torch.randint(0, 256, (n_img,), generator=gg).float().to(dev),
                torch.randn(n_gt, generator=gg).to(dev))

    a = frames(1)
    tab.set([a[0], a[1], a[2], None])
    ops.fetch_inputs(lib, tab.ptr, dst)
    backend.sync()
    assert torch.equal(dst[0].cpu(), a[0].cpu().float()) and torch.equal(dst[1].cpu(), a[1].cpu()) and torch.equal(dst[2].cpu(), a[2].cpu())
    assert torch.equal(dst[3].cpu(), keep3.cpu())                         # empty entry: untouched
    # the recorded op: one plan, the table rewritten between its runs; an entry naming the buffer itself is a no-op
    r = Recorder()
    ops.fetch_inputs(r, tab.ptr, dst)
    plan = r.compile()
    for seed in (2, 3):
        b = frames(seed)
        before1 = dst[1].clone()
        tab.set([b[0], dst[1], None, b[2]])
        plan.run(lib, backend.stream_handle() if hasattr(backend, "stream_handle") else 0)
        backend.sync()
        assert torch.equal(dst[0].cpu(), b[0].cpu().float()) and torch.equal(dst[1].cpu(), before1.cpu()) and torch.equal(dst[3].cpu(), b[2].cpu())
        assert torch.equal(dst[2].cpu(), a[2].cpu())                    # not named since the first call
    tab.clear()
    plan.run(lib, 0)
    backend.sync()
    assert torch.equal(dst[0].cpu(), b[0].cpu().float())
