"""Conv family parity: HIP kernels (or their CPU-emulated build) vs the torch oracle."""
import numpy as np
import pytest
import torch

from madnet_hip import ops
from oracle import tf_ops as T


def _rand(shape, seed, dev, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def _padded(x, ld):
    """copy [B,H,W,C] into a wider zero buffer [B,H,W,ld] and return (buffer, View)."""
    B, H, W, C = x.shape
    buf = torch.zeros(B, H, W, ld, device=x.device)
    buf[..., :C] = x
    return buf, ops.View(buf, B, H, W, C, ld)


# (B,H,W,Cin,Cout,k,stride,dil,in_ld_extra,alpha)
FWD_CASES = [
    (1, 6, 20, 8, 16, 3, 1, 1, 0, 0.2),        # tiny level-6 like
    (1, 9, 13, 3, 16, 3, 2, 1, 1, 0.2),        # conv1-like: Cin=3 padded to 4, stride 2, odd size
    (2, 8, 12, 16, 32, 3, 2, 1, 0, 0.2),       # batch 2, even size stride 2 (pad only bottom/right)
    (1, 7, 11, 38, 20, 3, 1, 1, 2, 0.2),       # odd Cin (38 in ld 40), Cout not multiple of 16
    (1, 10, 12, 12, 1, 3, 1, 1, 0, 1.0),       # Cout = 1 (disp6/context7), linear
    (1, 12, 16, 8, 24, 3, 1, 4, 0, 0.2),       # dilation 4
    (1, 6, 9, 8, 8, 3, 1, 16, 0, 0.2),         # dilation 16 > feature size
    (1, 8, 8, 5, 7, 5, 2, 1, 0, 0.1),          # 5x5 stride 2, unaligned channels (scalar path)
    (1, 6, 6, 4, 4, 7, 2, 1, 0, 0.1),          # 7x7 stride 2
    (1, 5, 7, 16, 8, 1, 1, 1, 0, 0.1),         # 1x1
]


@pytest.mark.parametrize("case", FWD_CASES)
def test_conv2d_fwd(backend, case):
    B, H, W, Ci, Co, k, s, d, extra, alpha = case
    dev = backend.device
    x = _rand((B, H, W, Ci), 1, dev)
    w = _rand((k, k, Ci, Co), 2, dev, 0.3)
    b = _rand((Co,), 3, dev)
    ref = T.conv2d(x.cpu(), w.cpu(), b.cpu(), stride=s, dilation=d, alpha=alpha)
    xb, xv = _padded(x, Ci + extra)
    if extra:
        xb[..., Ci:] = 7.0   # garbage in the channel padding must not leak
    out = torch.full(ref.shape, float("nan"), device=dev)
    ops.conv2d_fwd(backend.lib, xv, w, b, ops.view(out), stride=s, dil=d, alpha=alpha)
    backend.sync()
    err = (out.cpu() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("shape", [(1, 40, 50, 32, 32), (1, 24, 80, 96, 96), (1, 16, 24, 128, 128), (2, 12, 40, 70, 128)])
def test_conv2d_fwd_big_tiles(backend, shape):
    """Shapes that select the larger tile configurations (more than one M tile, N tile = Cout)."""
    if backend.name == "emul" and shape[3] * shape[4] > 96 * 96:
        pytest.skip("too slow on the CPU emulator")
    B, H, W, Ci, Co = shape
    dev = backend.device
    x = _rand((B, H, W, Ci), 4, dev)
    w = _rand((3, 3, Ci, Co), 5, dev, 0.1)
    b = _rand((Co,), 6, dev)
    ref = T.conv2d(x.cpu(), w.cpu(), b.cpu(), alpha=0.2)
    ld = (Ci + 3) // 4 * 4
    xb, xv = _padded(x, ld)
    out = torch.zeros_like(ref, device=dev)
    ops.conv2d_fwd(backend.lib, xv, w, b, ops.view(out), alpha=0.2)
    backend.sync()
    err = (out.cpu() - ref).abs().max().item()
    assert err <= 3e-5 * max(1.0, ref.abs().max().item()), err


DG_CASES = [
    (1, 6, 20, 8, 16, 3, 1, 1),
    (2, 8, 12, 16, 32, 3, 2, 1),       # stride-2 dgrad (lattice masks)
    (1, 9, 13, 3, 16, 3, 2, 1),        # odd size stride 2, Cin=3
    (1, 7, 11, 38, 20, 3, 1, 1),
    (1, 10, 12, 12, 1, 3, 1, 1),       # Cout=1: K=1 contraction (scalar weight path)
    (1, 12, 16, 8, 24, 3, 1, 4),
    (1, 8, 8, 5, 7, 5, 2, 1),
    (1, 16, 40, 8, 16, 3, 1, 1),       # 640 pixels: several pixel splits (atomics / partial-sum workspace)
]


def _oracle_grads(x, w, b, s, d, alpha, gy):
    x = x.clone().requires_grad_(True); w = w.clone().requires_grad_(True); b = b.clone().requires_grad_(True)
    y = T.conv2d(x, w, b, stride=s, dilation=d, alpha=alpha)
    gx, gw, gb = torch.autograd.grad(y, [x, w, b], gy)
    return y.detach(), gx, gw, gb


@pytest.mark.parametrize("case", DG_CASES)
def test_conv2d_dgrad_wgrad(backend, case):
    B, H, W, Ci, Co, k, s, d = case
    dev = backend.device
    x = _rand((B, H, W, Ci), 11, dev)
    w = _rand((k, k, Ci, Co), 12, dev, 0.3)
    b = _rand((Co,), 13, dev)
    Ho, Wo, _, _ = ops.conv_geometry(H, W, k, k, s, d)
    gz = _rand((B, Ho, Wo, Co), 14, dev)           # gradient w.r.t. the pre-activation (alpha=1 here)
    _, gx_ref, gw_ref, gb_ref = _oracle_grads(x.cpu(), w.cpu(), b.cpu(), s, d, 1.0, gz.cpu())
    ldx = (Ci + 3) // 4 * 4
    ldz = (Co + 3) // 4 * 4
    xb, xv = _padded(x, ldx)
    zb, zv = _padded(gz, ldz)
    # dgrad with accumulate + fused leaky mask:  dx = (old + dgrad) * (ref>0 ? 1 : 0.2)
    old = _rand((B, H, W, Ci), 15, dev)
    mref = _rand((B, H, W, Ci), 16, dev)
    dxb, dxv = _padded(old, ldx)
    mb, mv = _padded(mref, ldx)
    ops.conv2d_dgrad(backend.lib, zv, w, dxv, stride=s, dil=d, accumulate=True, mask_ref=mv, mask_alpha=0.2)
    dw = torch.zeros_like(w); db = torch.zeros_like(b)
    ops.conv2d_wgrad(backend.lib, xv, zv, dw, db, stride=s, dil=d)
    backend.sync()
    exp_dx = (old.cpu() + gx_ref) * torch.where(mref.cpu() > 0, 1.0, 0.2)
    sc = max(1.0, gx_ref.abs().max().item())
    assert (dxb[..., :Ci].cpu() - exp_dx).abs().max().item() <= 3e-5 * sc
    assert (dw.cpu() - gw_ref).abs().max().item() <= 1e-4 * max(1.0, gw_ref.abs().max().item())
    assert (db.cpu() - gb_ref).abs().max().item() <= 1e-4 * max(1.0, gb_ref.abs().max().item())
    # atomic-free form: per-split partial sums in a workspace + one reduction launch (dw2 is overwritten, not
    # accumulated: start from garbage)
    for prec in (0, 1):
        wsa = ops.WgradWorkspace(dev); wsa.CHUNK = 1 << 16
        segs, keep = [], []
        dw2 = torch.full_like(w, float("nan")); db2 = torch.zeros_like(b)
        ops.PRECISION = prec
        try:
            ops.conv2d_wgrad_partial(backend.lib, backend.lib, wsa, segs, xv, zv, dw2, db2, stride=s, dil=d)
        finally:
            ops.PRECISION = 0
        ops.wgrad_reduce(backend.lib, segs, dev, keep)
        backend.sync()
        tol = 1e-4 if prec == 0 else 2e-2
        assert (dw2.cpu() - gw_ref).abs().max().item() <= tol * max(1.0, gw_ref.abs().max().item()), prec
        assert (db2.cpu() - gb_ref).abs().max().item() <= 1e-4 * max(1.0, gb_ref.abs().max().item())


def test_conv2d_transpose(backend):
    dev = backend.device
    B, H, W, Ci, Co = 1, 5, 7, 8, 12
    x = _rand((B, H, W, Ci), 21, dev)
    w = _rand((4, 4, Co, Ci), 22, dev, 0.3)
    b = _rand((Co,), 23, dev)
    ref = T.conv2d_transpose(x.cpu(), w.cpu(), b.cpu(), stride=2, alpha=0.1)
    out = torch.zeros_like(ref, device=dev)
    ops.conv2d_transpose_fwd(backend.lib, ops.view(x), w, b, ops.view(out), stride=2, alpha=0.1)
    backend.sync()
    assert (out.cpu() - ref).abs().max().item() <= 3e-5 * max(1.0, ref.abs().max().item())


# forced tiles: (bm, bn, kt) x shapes whose K is a multiple of kt -> uniform-tap (UNI) loader fast path
UNI_CASES = [
    ((32, 32, 64), (1, 9, 14, 64, 32, 1, 2)),     # dilation 2
    ((64, 32, 64), (2, 8, 12, 128, 24, 1, 1)),
    ((128, 32, 32), (1, 12, 20, 32, 32, 2, 1)),   # stride 2 forward (dgrad falls back to the generic loader)
    ((64, 64, 32), (1, 10, 16, 96, 64, 1, 1)),
    ((32, 64, 128), (1, 6, 10, 128, 40, 1, 4)),
]


@pytest.mark.parametrize("tile,shape", UNI_CASES)
def test_conv_uniform_tap_fast_path(backend, tile, shape):
    bm, bn, kt = tile
    B, H, W, Ci, Co, s, d = shape
    dev = backend.device
    x = _rand((B, H, W, Ci), 31, dev)
    w = _rand((3, 3, Ci, Co), 32, dev, 0.2)
    b = _rand((Co,), 33, dev)
    Ho, Wo, _, _ = ops.conv_geometry(H, W, 3, 3, s, d)
    gz = _rand((B, Ho, Wo, Co), 34, dev)
    y_ref, gx_ref, _, _ = _oracle_grads(x.cpu(), w.cpu(), b.cpu(), s, d, 1.0, gz.cpu())
    outs = {}
    for no_uni in (0, 1):
        backend.lib.tune_conv_tile(bm | (no_uni << 16), bn | (kt << 16))
        try:
            y = torch.full(y_ref.shape, float("nan"), device=dev)
            ops.conv2d_fwd(backend.lib, ops.view(x), w, b, ops.view(y), stride=s, dil=d, alpha=1.0)
            dx = torch.full(x.shape, float("nan"), device=dev)
            ops.conv2d_dgrad(backend.lib, ops.view(gz), w, ops.view(dx), stride=s, dil=d)
            backend.sync()
        finally:
            backend.lib.tune_conv_tile(0, 0)
        assert (y.cpu() - y_ref).abs().max().item() <= 3e-5 * max(1.0, y_ref.abs().max().item())
        assert (dx.cpu() - gx_ref).abs().max().item() <= 3e-5 * max(1.0, gx_ref.abs().max().item())
        outs[no_uni] = (y.cpu(), dx.cpu())
    # same K order in both loaders -> bitwise identical results
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


# bf16 forward layers whose K is NOT a multiple of the K-tile (DispNet's iconv layers: 1024+1 / 768+1 / 384+1 concat channels): uniform-tap walk over K
# rounded up to whole tiles (conv_igemm_kernel<..., RAG>); (tile, (B, H, W, Ci, in_ld, Co, stride, dil))
RAGGED_CASES = [
    ((32, 64, 64), (1, 7, 10, 257, 260, 64, 1, 1)),        # one real channel in the last tile (split-K groups: 37 K-tiles on 4 workgroups)
    ((32, 32, 64), (1, 6, 9, 321, 324, 40, 1, 1)),         # output columns not a multiple of the tile
    ((64, 64, 64), (2, 5, 8, 300, 300, 64, 1, 2)),         # K % 4 == 0, row stride == K (the groups behind K do not exist), dilation 2
    ((64, 32, 64), (1, 9, 11, 385, 388, 32, 2, 1)),        # stride 2
    ((0, 0, 0), (1, 8, 12, 385, 388, 192, 1, 1)),          # tile picked by the heuristic
]


@pytest.mark.parametrize("tile,shape", RAGGED_CASES)
def test_conv_bf16_ragged_k_uniform_tap(backend, tile, shape):
    bm, bn, kt = tile
    B, H, W, Ci, ld, Co, s, d = shape
    dev = backend.device
    x = _rand((B, H, W, Ci), 61, dev)
    w = _rand((3, 3, Ci, Co), 62, dev, 0.1)
    b = _rand((Co,), 63, dev)
    y_ref = T.conv2d(_bf(x.cpu()), _bf(w.cpu()), b.cpu(), stride=s, dilation=d, alpha=0.2)
    xb, xv = _padded(x, ld)
    if ld > Ci:
        xb[..., Ci:] = float("inf")       # the padding channels of a row must never reach an MFMA (Inf x a dropped weight row = NaN)
    outs = {}
    ops.PRECISION = 1
    try:
        for ragged in (1, 0):
            backend.lib.tune_conv_tile(bm | ((1 - ragged) << 19), bn | (kt << 16))
            y = torch.full(y_ref.shape, float("nan"), device=dev)
            ops.conv2d_fwd(backend.lib, xv, w, b, ops.view(y), stride=s, dil=d, alpha=0.2)
            name = backend.lib.last_kernel().decode()
            assert ("uni-ragged" in name) == bool(ragged) and ",bf16," in name, name
            backend.sync()
            outs[ragged] = y.cpu()
    finally:
        ops.PRECISION = 0
        backend.lib.tune_conv_tile(0, 0)
    tol = 1e-4 * max(1.0, y_ref.abs().max().item())
    assert (outs[1] - y_ref).abs().max().item() <= tol and (outs[0] - y_ref).abs().max().item() <= tol
    # 256 input channels and fewer stay on the generic loader (the padded walk would waste > 20 %)
    x2 = _rand((1, 6, 8, 130), 64, dev); x2b, x2v = _padded(x2, 132)
    ops.PRECISION = 1
    try:
        ops.conv2d_fwd(backend.lib, x2v, _rand((3, 3, 130, 32), 65, dev), None, ops.view(torch.empty(1, 6, 8, 32, device=dev)))
        assert "gen" in backend.lib.last_kernel().decode()
    finally:
        ops.PRECISION = 0



BF16_CASES = [(1, 9, 14, 64, 32, 1, 2), (2, 8, 12, 128, 24, 1, 1), (1, 12, 20, 32, 32, 2, 1), (1, 7, 11, 38, 20, 1, 1),
              (1, 10, 16, 96, 64, 1, 1), (1, 9, 13, 3, 16, 2, 1), (1, 6, 10, 128, 128, 1, 4)]


@pytest.mark.parametrize("shape", BF16_CASES)
def test_conv_bf16_throughput_mode(backend, shape):
    """precision=1: operands rounded to bf16 (RNE) at the LDS store, fp32 accumulation.  Reference = the fp32
    oracle applied to bf16-rounded operands (products of bf16 values are exact in fp32)."""
    B, H, W, Ci, Co, s, d = shape
    dev = backend.device
    x = _rand((B, H, W, Ci), 41, dev)
    w = _rand((3, 3, Ci, Co), 42, dev, 0.2)
    b = _rand((Co,), 43, dev)
    Ho, Wo, _, _ = ops.conv_geometry(H, W, 3, 3, s, d)
    gz = _rand((B, Ho, Wo, Co), 44, dev)
    y_ref = T.conv2d(_bf(x.cpu()), _bf(w.cpu()), b.cpu(), stride=s, dilation=d, alpha=0.2)
    _, gx_ref, _, _ = _oracle_grads(x.cpu(), _bf(w.cpu()), b.cpu(), s, d, 1.0, _bf(gz.cpu()))
    ld = (Ci + 3) // 4 * 4
    xb, xv = _padded(x, ld)
    ops.PRECISION = 1
    try:
        y = torch.full(y_ref.shape, float("nan"), device=dev)
        ops.conv2d_fwd(backend.lib, xv, w, b, ops.view(y), stride=s, dil=d, alpha=0.2)
        dxb, dxv = _padded(torch.full(x.shape, float("nan"), device=dev), ld)
        ops.conv2d_dgrad(backend.lib, ops.view(gz), w, dxv, stride=s, dil=d)
        dw = torch.zeros_like(w)
        db = torch.zeros_like(b)
        ops.conv2d_wgrad(backend.lib, xv, ops.view(gz), dw, db, stride=s, dil=d)
        backend.sync()
    finally:
        ops.PRECISION = 0
    # filter gradient: bf16-rounded activations x bf16-rounded dz, fp32 accumulate; bias gradient stays exact fp32
    _, _, gw_ref, _ = _oracle_grads(_bf(x.cpu()), w.cpu(), b.cpu(), s, d, 1.0, _bf(gz.cpu()))
    _, _, gw32, gb_ref = _oracle_grads(x.cpu(), w.cpu(), b.cpu(), s, d, 1.0, gz.cpu())
    assert (dw.cpu() - gw_ref).abs().max().item() <= 1e-4 * max(1.0, gw_ref.abs().max().item())
    assert (db.cpu() - gb_ref).abs().max().item() <= 1e-4 * max(1.0, gb_ref.abs().max().item())
    if Ci % 4 == 0 and Co % 4 == 0:
        assert (dw.cpu() - gw32).abs().max().item() > 1e-5      # the bf16 kernel really ran
    assert (y.cpu() - y_ref).abs().max().item() <= 1e-4 * max(1.0, y_ref.abs().max().item())
    assert (dxb[..., :Ci].cpu() - gx_ref).abs().max().item() <= 1e-4 * max(1.0, gx_ref.abs().max().item())
    # and it really is a reduced-precision path: it differs from the exact fp32 result
    y32 = T.conv2d(x.cpu(), w.cpu(), b.cpu(), stride=s, dilation=d, alpha=0.2)
    assert (y.cpu() - y32).abs().max().item() > 1e-4


# bf16 mode with forced BIG tiles (wave-specialised kernels: 4 MFMA waves + 4 loader waves per workgroup)
WS_CASES = [((128, 64), (1, 10, 16, 64, 64, 1, 1)), ((128, 128), (1, 9, 15, 128, 128, 1, 2)), ((64, 64), (2, 8, 12, 96, 40, 1, 1)),
            ((64, 128), (1, 12, 20, 32, 128, 2, 1)), ((128, 32), (1, 7, 11, 38, 20, 1, 1)), ((128, 96), (1, 6, 10, 64, 96, 1, 1))]


@pytest.mark.parametrize("tile,shape", WS_CASES)
def test_conv_bf16_big_tiles(backend, tile, shape):
    bm, bn = tile
    B, H, W, Ci, Co, s, d = shape
    dev = backend.device
    x = _rand((B, H, W, Ci), 61, dev)
    w = _rand((3, 3, Ci, Co), 62, dev, 0.2)
    b = _rand((Co,), 63, dev)
    Ho, Wo, _, _ = ops.conv_geometry(H, W, 3, 3, s, d)
    gz = _rand((B, Ho, Wo, Co), 64, dev)
    y_ref = T.conv2d(_bf(x.cpu()), _bf(w.cpu()), b.cpu(), stride=s, dilation=d, alpha=0.2)
    _, gx_ref, _, _ = _oracle_grads(x.cpu(), _bf(w.cpu()), b.cpu(), s, d, 1.0, _bf(gz.cpu()))
    ld = (Ci + 3) // 4 * 4
    xb, xv = _padded(x, ld)
    ops.PRECISION = 1
    try:
        backend.lib.tune_conv_tile(bm, bn | (64 << 16))
        y = torch.full(y_ref.shape, float("nan"), device=dev)
        ops.conv2d_fwd(backend.lib, xv, w, b, ops.view(y), stride=s, dil=d, alpha=0.2)
        backend.lib.tune_conv_tile(bm, (128 if (bn > Ci and Ci > 64) else bn) | (64 << 16))
        dxb, dxv = _padded(torch.full(x.shape, float("nan"), device=dev), ld)
        ops.conv2d_dgrad(backend.lib, ops.view(gz), w, dxv, stride=s, dil=d)
        backend.sync()
    finally:
        ops.PRECISION = 0
        backend.lib.tune_conv_tile(0, 0)
    assert (y.cpu() - y_ref).abs().max().item() <= 1e-4 * max(1.0, y_ref.abs().max().item())
    assert (dxb[..., :Ci].cpu() - gx_ref).abs().max().item() <= 1e-4 * max(1.0, gx_ref.abs().max().item())


THIN_CASES = [(2, 12, 20, 16, 16, 1), (1, 13, 17, 3, 16, 2), (1, 12, 16, 16, 32, 2), (1, 9, 14, 32, 32, 1), (1, 10, 12, 8, 16, 1)]


@pytest.mark.parametrize("case", THIN_CASES)
def test_conv_thin_layers(backend, case):
    """bf16 weights-stationary kernel of the thin full-resolution layers (3x3, Cin <= 32, Cout 16/32): forward (stride 1 / 2)
    with bias + leaky, and the stride-1 input gradient with accumulate + fused leaky-gradient mask."""
    B, H, W, Ci, Co, s = case
    dev = backend.device
    x = _rand((B, H, W, Ci), 71, dev)
    w = _rand((3, 3, Ci, Co), 72, dev, 0.2)
    b = _rand((Co,), 73, dev)
    Ho, Wo, _, _ = ops.conv_geometry(H, W, 3, 3, s, 1)
    gz = _rand((B, Ho, Wo, Co), 74, dev)
    y_ref = T.conv2d(_bf(x.cpu()), _bf(w.cpu()), b.cpu(), stride=s, dilation=1, alpha=0.2)
    _, gx_ref, _, _ = _oracle_grads(x.cpu(), _bf(w.cpu()), b.cpu(), s, 1, 1.0, _bf(gz.cpu()))
    ld = (Ci + 3) // 4 * 4
    xb, xv = _padded(x, ld)
    if ld != Ci:
        xb[..., Ci:] = float("nan")                      # channel padding must never leak into the result
    old = _rand((B, H, W, Ci), 75, dev); mref = _rand((B, H, W, Ci), 76, dev)
    dxb, dxv = _padded(old, ld); mb, mv = _padded(mref, ld)
    ops.PRECISION = 1
    backend.lib.tune_conv_thin(1)
    try:
        y = torch.full(y_ref.shape, float("nan"), device=dev)
        ops.conv2d_fwd(backend.lib, xv, w, b, ops.view(y), stride=s, alpha=0.2)
        if s == 1:
            ops.conv2d_dgrad(backend.lib, ops.view(gz), w, dxv, stride=1, accumulate=True, mask_ref=mv, mask_alpha=0.2)
        backend.sync()
    finally:
        ops.PRECISION = 0
        backend.lib.tune_conv_thin(0)
    assert (y.cpu() - y_ref).abs().max().item() <= 1e-4 * max(1.0, y_ref.abs().max().item())
    if s == 1:
        exp = (old.cpu() + gx_ref) * torch.where(mref.cpu() > 0, 1.0, 0.2)
        assert (dxb[..., :Ci].cpu() - exp).abs().max().item() <= 1e-4 * max(1.0, gx_ref.abs().max().item())


# (B, H, W, Cin, Cout, dilation, mh_tune_conv_patch mode): 128 / 64 = pixel tile, +256 = 8-wave variant (K = 64 / 128 take its
# compile-time-K instances for the forward pass, Cout = 64 / 128 for the gradient), +2048 = generic instances only
PATCH_CASES = [(1, 12, 20, 128, 128, 1, 128), (1, 9, 21, 96, 64, 2, 64), (2, 10, 18, 38, 128, 1, 64), (1, 14, 19, 64, 96, 4, 128),
               (1, 8, 16, 128, 96, 1, 128 + 256), (1, 11, 17, 32, 64, 2, 1), (1, 7, 33, 160, 128, 1, 64),
               (1, 17, 35, 128, 128, 2, 128 + 256), (2, 9, 18, 64, 64, 1, 128 + 256), (1, 10, 20, 64, 128, 1, 128 + 256),
               (1, 12, 19, 96, 128, 1, 128 + 256), (1, 9, 17, 128, 64, 1, 128 + 256 + 2048)]


# K = 32 (one chunk per tap, five weight tiles with a dead last half) in both directions.  Added after the round's GPU budget
# was spent: run on the emulator only (the same instances run on the MI355X inside the engine / bench: the 64->32 layer's dgrad)
PATCH_CASES_K32 = [(1, 9, 17, 64, 32, 1, 128 + 256), (1, 9, 17, 32, 64, 1, 128 + 256), (1, 9, 18, 32, 48, 2, 64),
                   (1, 9, 17, 64, 160, 1, 128 + 256)]      # + Cout = 160: two column tiles, the second one partly dead (no such layer in either net)
# 32-column tile (round 3): the layers with <= 32 output columns -- 32 -> 32, 64 -> 32 and their input gradients
PATCH_CASES_THIN = [(2, 9, 21, 32, 32, 1, 128 + 256), (1, 10, 18, 64, 32, 2, 64), (1, 9, 33, 32, 64, 1, 128 + 256), (1, 11, 19, 40, 36, 1, 128)]


@pytest.mark.parametrize("case", PATCH_CASES + PATCH_CASES_K32 + PATCH_CASES_THIN)
def test_conv_bf16_patch_kernel(backend, case):
    """Patch-staged bf16 kernel of the stride-1 3x3 (dilated) layers (csrc/conv_patch.hip): forward with bias + leaky and the
    input gradient with accumulate + leaky-grad mask, against the oracle on bf16-rounded operands; every pixel tile
    (64 / 128 pixels, 8-wave variant), dilation sub-lattices with ragged edges, Cin = 38 (row padding must not leak)."""
    B, H, W, Ci, Co, dil, mode = case
    if case in PATCH_CASES_K32 and backend.name != "emul":
        pytest.skip("emulator-only case (see PATCH_CASES_K32)")
    dev = backend.device
    x = _rand((B, H, W, Ci), 91, dev)
    w = _rand((3, 3, Ci, Co), 92, dev, 0.2)
    b = _rand((Co,), 93, dev)
    gz = _rand((B, H, W, Co), 94, dev)
    y_ref = T.conv2d(_bf(x.cpu()), _bf(w.cpu()), b.cpu(), stride=1, dilation=dil, alpha=0.2)
    _, gx_ref, _, _ = _oracle_grads(x.cpu(), _bf(w.cpu()), b.cpu(), 1, dil, 1.0, _bf(gz.cpu()))
    ld = (Ci + 3) // 4 * 4
    xb, xv = _padded(x, ld)
    if ld != Ci:
        xb[..., Ci:] = float("nan")
    old = _rand((B, H, W, Ci), 95, dev); mref = _rand((B, H, W, Ci), 96, dev)
    dxb, dxv = _padded(old, ld); mb, mv = _padded(mref, ld)
    ops.PRECISION = 1
    backend.lib.tune_conv_patch(mode)
    try:
        y = torch.full(y_ref.shape, float("nan"), device=dev)
        ops.conv2d_fwd(backend.lib, xv, w, b, ops.view(y), stride=1, dil=dil, alpha=0.2)
        ops.conv2d_dgrad(backend.lib, ops.view(gz), w, dxv, stride=1, dil=dil, accumulate=True, mask_ref=mv, mask_alpha=0.2)
        backend.sync()
    finally:
        ops.PRECISION = 0
        launches = backend.lib.tune_conv_patch(-1)
    # dispatch rule (mh_conv_patch_ok): >= 32 output columns (a 32-column tile since round 3), >= 32 reduction channels, 16-byte rows;
    # mode 1: these shapes are too small for the tile heuristic -> gather kernel
    fwd_ok = Co >= 32 and Ci >= 32 and Co % 4 == 0
    dgrad_ok = Ci >= 32 and Co >= 32            # (the gradient rows are padded to a multiple of 4 floats here: Cin = 38 qualifies since round 3, #18)
    assert launches == (0 if mode == 1 else int(fwd_ok) + int(dgrad_ok))
    assert (y.cpu() - y_ref).abs().max().item() <= 1e-4 * max(1.0, y_ref.abs().max().item())
    exp = (old.cpu() + gx_ref) * torch.where(mref.cpu() > 0, 1.0, 0.2)
    assert (dxb[..., :Ci].cpu() - exp).abs().max().item() <= 1e-4 * max(1.0, gx_ref.abs().max().item())


@pytest.mark.parametrize("case", [(1, 10, 18, 128, 128, 2, 128 + 256), (2, 7, 16, 96, 64, 1, 64)])
def test_conv_bf16_patch_kernel_plain_epilogue(backend, case):
    """Same kernel with the other epilogue combinations: forward without bias / activation, input gradient overwriting its
    destination (no accumulate, no mask)."""
    B, H, W, Ci, Co, dil, mode = case
    dev = backend.device
    x = _rand((B, H, W, Ci), 101, dev)
    w = _rand((3, 3, Ci, Co), 102, dev, 0.2)
    gz = _rand((B, H, W, Co), 103, dev)
    zero_b = torch.zeros(Co)
    y_ref = T.conv2d(_bf(x.cpu()), _bf(w.cpu()), zero_b, stride=1, dilation=dil, alpha=1.0)
    _, gx_ref, _, _ = _oracle_grads(x.cpu(), _bf(w.cpu()), zero_b, 1, dil, 1.0, _bf(gz.cpu()))
    ops.PRECISION = 1
    backend.lib.tune_conv_patch(mode)
    try:
        y = torch.full(y_ref.shape, float("nan"), device=dev)
        dx = torch.full((B, H, W, Ci), float("nan"), device=dev)
        ops.conv2d_fwd(backend.lib, ops.view(x), w, None, ops.view(y), stride=1, dil=dil, alpha=1.0)
        ops.conv2d_dgrad(backend.lib, ops.view(gz), w, ops.view(dx), stride=1, dil=dil)
        backend.sync()
    finally:
        ops.PRECISION = 0
        launches = backend.lib.tune_conv_patch(-1)
    assert launches == 2
    assert (y.cpu() - y_ref).abs().max().item() <= 1e-4 * max(1.0, y_ref.abs().max().item())
    assert (dx.cpu() - gx_ref).abs().max().item() <= 1e-4 * max(1.0, gx_ref.abs().max().item())


@pytest.mark.parametrize("case", [(1, 20, 28, 3, 16, 7, 2), (2, 11, 13, 3, 32, 5, 1), (1, 10, 12, 4, 64, 3, 1), (1, 9, 9, 1, 16, 3, 1)])
def test_wgrad_bf16_tap_flattened(backend, case):
    """Cin <= 4 in bf16 mode: the filter-gradient tile rows are (tap, channel) pairs, BK/4 taps per workgroup share one dz
    tile (DispNet's 7x7 image layer, MADNet's 3x3 one); compared with the oracle on bf16-rounded operands."""
    B, H, W, Ci, Co, k, s = case
    dev = backend.device
    x = _rand((B, H, W, Ci), 81, dev)
    w = _rand((k, k, Ci, Co), 82, dev, 0.2)
    b = _rand((Co,), 83, dev)
    Ho, Wo, _, _ = ops.conv_geometry(H, W, k, k, s, 1)
    gz = _rand((B, Ho, Wo, Co), 84, dev)
    _, _, gw_ref, _ = _oracle_grads(_bf(x.cpu()), w.cpu(), b.cpu(), s, 1, 1.0, _bf(gz.cpu()))
    _, _, _, gb_ref = _oracle_grads(x.cpu(), w.cpu(), b.cpu(), s, 1, 1.0, gz.cpu())
    ld = (Ci + 3) // 4 * 4
    xb, xv = _padded(x, ld)
    if ld != Ci:
        xb[..., Ci:] = 7.0                               # junk in the channel padding must not reach dw
    ops.PRECISION = 1
    try:
        dw = torch.zeros_like(w); db = torch.zeros_like(b)
        ops.conv2d_wgrad(backend.lib, xv, ops.view(gz), dw, db, stride=s)
        wsa = ops.WgradWorkspace(dev); wsa.CHUNK = 1 << 16
        segs, keep = [], []
        dw2 = torch.full_like(w, float("nan")); db2 = torch.zeros_like(b)
        ops.conv2d_wgrad_partial(backend.lib, backend.lib, wsa, segs, xv, ops.view(gz), dw2, db2, stride=s)
        ops.wgrad_reduce(backend.lib, segs, dev, keep)
        backend.sync()
    finally:
        ops.PRECISION = 0
    tol = 1e-4 * max(1.0, gw_ref.abs().max().item())
    assert (dw.cpu() - gw_ref).abs().max().item() <= tol
    assert (dw2.cpu() - gw_ref).abs().max().item() <= tol
    assert (db.cpu() - gb_ref).abs().max().item() <= 1e-4 * max(1.0, gb_ref.abs().max().item())


# (B, H, W, Cin, Cout, dilation): Cout > 64 -> the 64-pixel x 128-column instance, Cout <= 64 -> 128 pixels x 64 columns; odd chunk
# counts (Cin = 96: 27 chunks), Cin = 38 (row padding), ragged dilation lattices, two column tiles (Cout = 160)
X3_CASES = [(1, 12, 20, 128, 128, 1), (1, 9, 21, 96, 64, 2), (2, 10, 18, 38, 128, 1), (1, 14, 19, 64, 96, 4), (1, 7, 33, 128, 96, 8),
            (1, 9, 17, 32, 64, 1), (1, 9, 17, 64, 160, 1), (1, 11, 35, 36, 128, 1)]


@pytest.mark.parametrize("generic", [False, True], ids=["ctk", "generic"])
@pytest.mark.parametrize("case", X3_CASES)
def test_conv_split_bf16_patch_kernel(backend, case, generic):
    """precision code 2 (split-bf16, three bf16 MFMAs per product) on the patch-staged forward kernel: judged against the
    UNROUNDED fp32 oracle -- the error must be ~2^-16 relative (>= 50x below plain bf16), which is what lets the forward pass
    of the 'mixed' engine mode stay inside the 1e-3 px tolerance while the matrix cores run bf16."""
    B, H, W, Ci, Co, dil = case
    if generic and Ci not in (64, 128):
        pytest.skip("only K = 64 / 128 have compile-time-K instances to switch off")
    dev = backend.device
    x = _rand((B, H, W, Ci), 111, dev)
    w = _rand((3, 3, Ci, Co), 112, dev, 0.2)
    b = _rand((Co,), 113, dev)
    y_ref = T.conv2d(x.cpu().double(), w.cpu().double(), b.cpu().double(), stride=1, dilation=dil, alpha=0.2).float()
    y_bf = T.conv2d(_bf(x.cpu()), _bf(w.cpu()), b.cpu(), stride=1, dilation=dil, alpha=0.2)
    ld = (Ci + 3) // 4 * 4
    xb, xv = _padded(x, ld)
    if ld != Ci:
        xb[..., Ci:] = float("nan")
    backend.lib.tune_conv_patch(128 + (2048 if generic else 0))   # forced: these shapes are far below the pixel-count heuristic; +2048 = generic-K instances only
    try:
        y = torch.full(y_ref.shape, float("nan"), device=dev)
        with ops.precision_scope("mixed"):
            ops.conv2d_fwd(backend.lib, xv, w, b, ops.view(y), stride=1, dil=dil, alpha=0.2)
        backend.sync()
    finally:
        launches = backend.lib.tune_conv_patch(-1)
    assert launches == 1
    err = (y.cpu() - y_ref).abs().max().item()
    err_bf = (y_bf - y_ref).abs().max().item()
    scale = max(1.0, y_ref.abs().max().item())
    assert err <= 4e-5 * scale, (err, err_bf)
    assert err * 50 <= err_bf, (err, err_bf)


def test_precision_code_2_falls_back_to_exact_fp32(backend):
    """Layers without an x3 instance (stride 2, 5x5, few output channels) and every gradient run exact fp32 for code 2."""
    dev = backend.device
    x = _rand((1, 9, 14, 16), 121, dev); w = _rand((3, 3, 16, 32), 122, dev, 0.2); b = _rand((32,), 123, dev)
    y0 = torch.empty(1, 5, 7, 32, device=dev); y2 = torch.empty_like(y0)
    ops.conv2d_fwd(backend.lib, ops.view(x), w, b, ops.view(y0), stride=2, alpha=0.2)
    ops.PRECISION = 2
    try:
        ops.conv2d_fwd(backend.lib, ops.view(x), w, b, ops.view(y2), stride=2, alpha=0.2)
    finally:
        ops.PRECISION = 0
    backend.sync()
    assert torch.equal(y0.cpu(), y2.cpu())


# (B, H, W, Cin, Cout, stride, dilation): the coarse estimator layers (Cin = 200 / 136 / 104 incl. a 32-k tail), pyramid layers (stride 1 / 2),
# ragged pixel counts, Cout = 1-wide tiles excluded (N >= 16)
GROUP_LAYERS = [   # (H, W, Cin, Cout, stride, dil, precision): one batch = the filter gradients a pyramid level issues together
    (12, 20, 128, 128, 1, 1, 1), (12, 20, 128, 96, 1, 1, 1), (12, 20, 96, 64, 1, 1, 1), (12, 20, 64, 32, 1, 1, 1), (12, 20, 32, 1, 1, 1, 1),
    (12, 20, 38, 128, 1, 2, 1), (12, 20, 16, 16, 2, 1, 1), (12, 20, 3, 16, 2, 1, 1), (12, 20, 16, 32, 1, 1, 0), (12, 20, 32, 64, 1, 1, 1),
]


def test_wgrad_partial_group_matches_single_launches(backend):
    """mh_conv2d_wgrad_partial_group (what mh_plan_run issues for consecutive partial-filter-gradient ops of one lane): the same layers in
    ONE grid must give bit-identical partial sums to one launch per layer (the per-workgroup arithmetic is the same code); bias
    gradients are atomics across workgroups -> compared with a tolerance.  10 layers = two grids of <= 8 + an exact-fp32 layer
    and leftovers that go out alone."""
    from madnet_hip import plan as PL
    dev = backend.device
    res = {}
    for how in ("single", "plan"):
        wsa = ops.WgradWorkspace(dev); wsa.CHUNK = 1 << 20
        rec = PL.Recorder() if how == "plan" else None
        tgt = rec if rec is not None else backend.lib
        segs, keep, outs = [], [], []
        for li, (H, W, Ci, Co, s, d, prec) in enumerate(GROUP_LAYERS):
            x = _rand((1, H, W, Ci), 100 + li, dev)
            Ho, Wo, _, _ = ops.conv_geometry(H, W, 3, 3, s, d)
            gz = _rand((1, Ho, Wo, Co), 200 + li, dev)
            xb, xv = _padded(x, (Ci + 3) // 4 * 4)
            zb, zv = _padded(gz, (Co + 3) // 4 * 4)
            dw = torch.full((3, 3, Ci, Co), float("nan"), device=dev); db = torch.zeros(Co, device=dev)
            ops.PRECISION_BWD = prec
            ops.PRECISION = prec
            try:
                ops.conv2d_wgrad_partial(tgt, backend.lib, wsa, segs, xv, zv, dw, db, stride=s, dil=d)
            finally:
                ops.PRECISION = 0
                ops.PRECISION_BWD = None
            outs.append((dw, db)); keep += [xb, zb]
        ops.wgrad_reduce(tgt, segs, dev, keep)
        if rec is not None:
            pl = rec.compile()
            assert pl.n == len(GROUP_LAYERS) + 1
            pl.run(backend.lib, None)
        backend.sync()
        res[how] = [(a.cpu().clone(), b.cpu().clone()) for a, b in outs]
    for li, ((dw1, db1), (dw2, db2)) in enumerate(zip(res["single"], res["plan"])):
        assert torch.isfinite(dw1).all() and torch.isfinite(dw2).all(), li
        assert torch.equal(dw1, dw2), (li, (dw1 - dw2).abs().max().item())
        assert (db1 - db2).abs().max().item() <= 1e-4 * max(1.0, db1.abs().max().item()), li


@pytest.mark.parametrize("case", X3_CASES + [(1, 12, 20, 64, 32, 1), (1, 9, 21, 40, 36, 2),
                                            (2, 9, 21, 32, 32, 1), (1, 10, 18, 64, 32, 2)])      # the 32-column tile
@pytest.mark.parametrize("small_tile", [False, True], ids=["tile128", "tile64x64"])
def test_conv_split_bf16_fragment_bank_kernel(backend, case, small_tile):
    """mh_conv2d_wb: the split-bf16 forward kernel that streams its weight operand from the MFMA fragment bank mh_pack_weights writes
    (no LDS staging of the weights, no barrier in the K walk).  Same arithmetic, same summation order as the LDS-staged split-bf16
    kernel -> compared bit for bit with it, and against the unrounded fp64 oracle at the 2^-16 level."""
    B, H, W, Ci, Co, dil = case
    dev = backend.device
    x = _rand((B, H, W, Ci), 111, dev)
    w = _rand((3, 3, Ci, Co), 112, dev, 0.2)
    b = _rand((Co,), 113, dev)
    y_ref = T.conv2d(x.cpu().double(), w.cpu().double(), b.cpu().double(), stride=1, dilation=dil, alpha=0.2).float()
    ld = (Ci + 3) // 4 * 4
    xb, xv = _padded(x, ld)
    if ld != Ci:
        xb[..., Ci:] = float("nan")
    bank = torch.full((ops.pack_bytes(w) // 4,), float("nan"), device=dev)
    assert ops.pack_bytes(w) == backend.lib.pack_bytes(9, Ci, Co, 2)
    keep = []
    ops.pack_weights(backend.lib, [(w, bank)], dev, keep)
    backend.lib.tune_conv_patch(128)          # forced: these shapes are far below the pixel-count heuristic
    backend.lib.tune_conv_bank(0)             # 0: the small-layer bank kernel stays out of the way (it has its own test)
    # small_tile: the 64-pixel x 64-column instance with 4 waves that under-filled grids take (1 << 30: always; 0: never)
    prev_tile = backend.lib.tune_conv_bank_tile((1 << 30) if small_tile else 0)
    try:
        y = torch.full(y_ref.shape, float("nan"), device=dev)
        y0 = torch.full(y_ref.shape, float("nan"), device=dev)
        with ops.precision_scope("mixed"):
            ops.conv2d_fwd(backend.lib, xv, w, b, ops.view(y), stride=1, dil=dil, alpha=0.2, wb=bank)
            name = backend.lib.last_kernel().decode()
            assert ("conv_bank_kernel<2,2,2,2" in name) == (small_tile and Co > 32), name        # (<= 32 columns: the 128x32 tile either way)
            nb = backend.lib.tune_conv_bank(0)
            ops.conv2d_fwd(backend.lib, xv, w, b, ops.view(y0), stride=1, dil=dil, alpha=0.2)
        backend.sync()
    finally:
        launches = backend.lib.tune_conv_patch(-1)
        nb2 = backend.lib.tune_conv_bank(-1)
        backend.lib.tune_conv_bank_tile(prev_tile)
    err = (y.cpu() - y_ref).abs().max().item()
    assert err <= 4e-5 * max(1.0, y_ref.abs().max().item()), err
    if Co >= 48:
        assert nb == 1 and launches == 2 and nb2 == 0
        assert torch.equal(y.cpu(), y0.cpu()), (y.cpu() - y0.cpu()).abs().max().item()
    else:       # 32..47 output channels: only the bank kernel has a split-bf16 instance (the LDS-staged one starts at 48; without a bank: exact fp32)
        assert nb == 1 and launches == 1 and nb2 == 0
        assert (y.cpu() - y0.cpu()).abs().max().item() <= 4e-5 * max(1.0, y_ref.abs().max().item())


SMALL_BANK_CASES = [   # (B, H, W, Cin, Cout, dil): the 1/16-1/64 level shapes + ragged ones
    (1, 6, 20, 128, 128, 1), (1, 6, 20, 197, 128, 1), (1, 12, 40, 128, 96, 1), (1, 7, 19, 96, 64, 1), (2, 6, 20, 64, 32, 1),
    (1, 9, 33, 134, 128, 1), (1, 12, 20, 32, 48, 2), (1, 5, 17, 36, 20, 1),
]


@pytest.mark.parametrize("x3", [False, True], ids=["bf16", "x3"])
@pytest.mark.parametrize("case", [(2, 24, 80, 64, 96), (2, 12, 40, 96, 128), (1, 13, 41, 128, 192), (1, 9, 20, 70, 48), (1, 48, 64, 32, 64)])
def test_conv_small_layer_bank_kernel_stride2(backend, case, x3):
    """The stride-2 forward instances of the small-layer bank kernel (pyramid conv7 / conv9 / conv11: 5 x 33 input patch per 2 x 16 output tile), even
    and odd input sizes (TF 'SAME': pad 0 / 1 in front), against the oracle on identically rounded operands."""
    B, H, W, Ci, Co = case
    dev = backend.device
    x = _rand((B, H, W, Ci), 711, dev)
    w = _rand((3, 3, Ci, Co), 712, dev, 0.2)
    b = _rand((Co,), 713, dev)
    ld = (Ci + 3) // 4 * 4
    xb, xv = _padded(x, ld)
    if ld != Ci:
        xb[..., Ci:] = float("nan")
    planes = 2 if x3 else 1
    keep = []
    bank = torch.full((ops.pack_bytes(w, planes) // 4,), float("nan"), device=dev)
    ops.pack_weights(backend.lib, [(w, bank, planes, 0)], dev, keep)
    Ho, Wo, _, _ = ops.conv_geometry(H, W, 3, 3, 2, 1)
    y = torch.full((B, Ho, Wo, Co), float("nan"), device=dev)
    backend.lib.tune_conv_bank(-1)
    with ops.precision_scope("mixed" if x3 else "bf16"):
        ops.conv2d_fwd(backend.lib, xv, w, b, ops.view(y), stride=2, alpha=0.2, wb=bank)
    name = backend.lib.last_kernel().decode()
    backend.sync()
    assert backend.lib.tune_conv_bank(-1) == 1 and "conv_bank_small" in name, name
    if x3:
        ref = T.conv2d(x.cpu().double(), w.cpu().double(), b.cpu().double(), stride=2, alpha=0.2).float(); tol = 4e-5
    else:
        ref = T.conv2d(_bf(x.cpu()).double(), _bf(w.cpu()).double(), b.cpu().double(), stride=2, alpha=0.2).float(); tol = 2e-5
    assert (y.cpu() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("what", ["fwd-bf16", "fwd-x3", "dgrad-bf16"])
@pytest.mark.parametrize("case", SMALL_BANK_CASES)
def test_conv_small_layer_bank_kernel(backend, case, what):
    """conv_bank_small_kernel (mh_conv2d_wb on layers of <= 4096 output pixels): 16 waves split the reduction of one 32x32 tile and fetch
    their weight fragments from the bank.  Same operand rounding as the tiled kernels of the same precision code (bf16 / split-bf16), another
    summation order: compared with the oracle run on identically rounded operands, and with the kernel the layer takes without a bank."""
    B, H, W, Ci, Co, dil = case
    dev = backend.device
    w = _rand((3, 3, Ci, Co), 312, dev, 0.2)
    b = _rand((Co,), 313, dev)
    keep = []
    backend.lib.tune_conv_bank(-1)
    if what.startswith("fwd"):
        x = _rand((B, H, W, Ci), 311, dev)
        ld = (Ci + 3) // 4 * 4
        xb, xv = _padded(x, ld)
        if ld != Ci:
            xb[..., Ci:] = float("nan")
        x3 = what == "fwd-x3"
        planes = 2 if x3 else 1
        bank = torch.full((ops.pack_bytes(w, planes) // 4,), float("nan"), device=dev)
        ops.pack_weights(backend.lib, [(w, bank, planes, 0)], dev, keep)
        y = torch.full((B, H, W, Co), float("nan"), device=dev); y0 = torch.full((B, H, W, Co), float("nan"), device=dev)
        with ops.precision_scope("mixed" if x3 else "bf16"):
            ops.conv2d_fwd(backend.lib, xv, w, b, ops.view(y), dil=dil, alpha=0.2, wb=bank)
            name = backend.lib.last_kernel().decode()
            nb = backend.lib.tune_conv_bank(-1)
            ops.conv2d_fwd(backend.lib, xv, w, b, ops.view(y0), dil=dil, alpha=0.2)
        backend.sync()
        assert nb == 1 and "conv_bank_small" in name, name
        if x3:
            ref = T.conv2d(x.cpu().double(), w.cpu().double(), b.cpu().double(), dilation=dil, alpha=0.2).float()
            tol = 4e-5
        else:
            ref = T.conv2d(_bf(x.cpu()).double(), _bf(w.cpu()).double(), b.cpu().double(), dilation=dil, alpha=0.2).float()
            tol = 2e-5
        sc = max(1.0, ref.abs().max().item())
        assert (y.cpu() - ref).abs().max().item() <= tol * sc
        assert (y.cpu() - y0.cpu()).abs().max().item() <= 2 * tol * sc
    else:
        gz = _rand((B, H, W, Co), 314, dev)
        old = _rand((B, H, W, Ci), 315, dev)
        mref = _rand((B, H, W, Ci), 316, dev)
        ldx = (Ci + 3) // 4 * 4
        zb, zv = _padded(gz, (Co + 3) // 4 * 4)
        mb, mv = _padded(mref, ldx)
        bank = torch.full((ops.pack_bytes(w, 1, 1) // 4,), float("nan"), device=dev)
        ops.pack_weights(backend.lib, [(w, bank, 1, 1)], dev, keep)
        outs = []
        for use_bank in (True, False):
            dxb, dxv = _padded(old, ldx)
            ops.PRECISION_BWD = 1
            try:
                ops.conv2d_dgrad(backend.lib, zv, w, dxv, dil=dil, accumulate=True, mask_ref=mv, mask_alpha=0.2, wb=(bank if use_bank else None))
            finally:
                ops.PRECISION_BWD = None
            if use_bank:
                name = backend.lib.last_kernel().decode()
                assert backend.lib.tune_conv_bank(-1) == 1 and "conv_bank_small_kernel<dgrad" in name, name
            backend.sync()
            outs.append(dxb[..., :Ci].cpu().clone())
        xr = torch.zeros(B, H, W, Ci, dtype=torch.float64, requires_grad=True)
        yr = T.conv2d(xr, _bf(w.cpu()).double(), None, dilation=dil, alpha=1.0)
        (gx,) = torch.autograd.grad(yr, [xr], _bf(gz.cpu()).double())
        exp = ((old.cpu().double() + gx) * torch.where(mref.cpu() > 0, 1.0, 0.2)).float()
        sc = max(1.0, gx.abs().max().item())
        assert (outs[0] - exp).abs().max().item() <= 2e-5 * sc
        assert (outs[0] - outs[1]).abs().max().item() <= 4e-5 * sc


@pytest.mark.parametrize("case", [(1, 6, 20, 128, 128, 1), (1, 12, 40, 134, 128, 1), (2, 24, 80, 64, 96, 1), (1, 9, 33, 96, 192, 1), (1, 12, 20, 32, 48, 2), (1, 5, 17, 36, 20, 1)])
def test_conv_small_layer_bank_kernel_placement_modes(backend, case):
    """mh_tune_conv_bank_small (round 6): the workgroup -> (XCD, tile) map of the small-layer kernel -- every XCD with the pixel-major order (0), the
    column-major order where the model prefers it (1), the layer confined to the fewest XCDs (2), both (3 = the default's model) -- is a permutation of
    the same tiles (plus workgroups that exit at once): forward and input gradient are bit-identical in every mode, every output element is written."""
    B, H, W, Ci, Co, dil = case
    dev = backend.device
    w = _rand((3, 3, Ci, Co), 812, dev, 0.2)
    b = _rand((Co,), 813, dev)
    x = _rand((B, H, W, Ci), 811, dev)
    gz = _rand((B, H, W, Co), 814, dev)
    mref = _rand((B, H, W, Ci), 816, dev)
    keep = []
    ldx = (Ci + 3) // 4 * 4
    xb, xv = _padded(x, ldx)
    zb, zv = _padded(gz, (Co + 3) // 4 * 4)
    mb, mv = _padded(mref, ldx)
    bank = torch.full((ops.pack_bytes(w, 1) // 4,), float("nan"), device=dev)
    bank_t = torch.full((ops.pack_bytes(w, 1, 1) // 4,), float("nan"), device=dev)
    ops.pack_weights(backend.lib, [(w, bank, 1, 0), (w, bank_t, 1, 1)], dev, keep)
    backend.lib.tune_conv_bank(-1)
    res = []
    prev = backend.lib.tune_conv_bank_small(-1)
    try:
        for mode in (0, 1, 2, 3, -1):
            backend.lib.tune_conv_bank_small(mode)
            y = torch.full((B, H, W, Co), float("nan"), device=dev)
            dxb = torch.full((B, H, W, ldx), float("nan"), device=dev)
            with ops.precision_scope("bf16"):
                ops.conv2d_fwd(backend.lib, xv, w, b, ops.view(y), dil=dil, alpha=0.2, wb=bank)
                assert "conv_bank_small_kernel<fwd" in backend.lib.last_kernel().decode()
            ops.PRECISION_BWD = 1
            try:
                ops.conv2d_dgrad(backend.lib, zv, w, ops.View(dxb, B, H, W, Ci, ldx), dil=dil, mask_ref=mv, mask_alpha=0.2, wb=bank_t)
                assert "conv_bank_small_kernel<dgrad" in backend.lib.last_kernel().decode()
            finally:
                ops.PRECISION_BWD = None
            backend.sync()
            res.append((y.cpu().clone(), dxb[..., :Ci].cpu().clone()))
    finally:
        backend.lib.tune_conv_bank_small(prev)
    assert torch.isfinite(res[0][0]).all() and torch.isfinite(res[0][1]).all()
    for y, dx in res[1:]:
        assert torch.equal(y, res[0][0]) and torch.equal(dx, res[0][1])


def test_wgrad_bf16_eight_wave_tile(backend):
    """Filter gradient of a layer that is launched on its own (> 4096 reduction pixels) with > 64 input and output channels: the 128x128
    tile with 8 waves of 32x64 (wgrad_bf16_kernel<4,2,2,4>), partial sums + reduction, against the oracle on bf16-rounded operands."""
    B, H, W, Ci, Co = 1, 130, 130, 72, 80
    dev = backend.device
    x = _rand((B, H, W, Ci), 411, dev)
    gz = _rand((B, H, W, Co), 412, dev)
    xb, xv = _padded(x, Ci)
    zb, zv = _padded(gz, Co)
    dw = torch.full((3, 3, Ci, Co), float("nan"), device=dev); db = torch.zeros(Co, device=dev)
    wsa = ops.WgradWorkspace(dev); segs, keep = [], []
    ops.PRECISION = 1
    try:
        ops.conv2d_wgrad_partial(backend.lib, backend.lib, wsa, segs, xv, zv, dw, db)
        name = backend.lib.last_kernel().decode()
    finally:
        ops.PRECISION = 0
    ops.wgrad_reduce(backend.lib, segs, dev, keep)
    backend.sync()
    assert "wgrad_bf16_kernel<4,2,2,4>" in name, name
    xr = _bf(x.cpu()).double(); w0 = torch.zeros(3, 3, Ci, Co, dtype=torch.float64, requires_grad=True)
    y = T.conv2d(xr, w0, None, alpha=1.0)
    (gw,) = torch.autograd.grad(y, [w0], _bf(gz.cpu()).double())
    assert (dw.cpu().double() - gw).abs().max().item() <= 2e-5 * max(1.0, gw.abs().max().item())
    assert (db.cpu().double() - gz.cpu().double().sum((0, 1, 2))).abs().max().item() <= 1e-4 * max(1.0, gz.cpu().abs().sum((0, 1, 2)).max().item())


X3_IGEMM_CASES = [   # (B, H, W, Cin, Cout, k, stride, dil): layers no patch / bank kernel takes -- strided, thin, wide-K, 5x5 / 7x7
    (2, 24, 40, 3, 16, 3, 2, 1), (2, 24, 40, 16, 16, 3, 1, 1), (1, 20, 36, 32, 64, 3, 2, 1), (1, 13, 29, 64, 96, 3, 2, 1), (1, 8, 20, 128, 192, 3, 2, 1),
    (1, 16, 24, 3, 64, 7, 2, 1), (1, 12, 20, 64, 128, 5, 2, 1), (1, 10, 16, 145, 256, 5, 2, 1), (1, 6, 10, 512, 512, 3, 1, 1), (1, 9, 17, 96, 64, 3, 1, 16),
    (1, 6, 20, 40, 20, 3, 1, 1),
]


@pytest.mark.parametrize("case", X3_IGEMM_CASES)
def test_conv_split_bf16_tiled_kernel(backend, case):
    """precision code 2 on the tiled implicit-GEMM kernel (conv_igemm_kernel<..., X3>): hi / lo planes of both tiles in LDS, three bf16 MFMAs per
    product -- the forward layers that have no patch / bank instance (the MADNet pyramid, DispNet's strided 7x7 / 5x5 / wide layers) judged
    against the UNROUNDED fp64 oracle at the 2^-16 level, bias + leaky + accumulate epilogue included."""
    B, H, W, Ci, Co, k, s, d = case
    dev = backend.device
    x = _rand((B, H, W, Ci), 611, dev)
    w = _rand((k, k, Ci, Co), 612, dev, 0.2)
    b = _rand((Co,), 613, dev)
    ref = T.conv2d(x.cpu().double(), w.cpu().double(), b.cpu().double(), stride=s, dilation=d, alpha=0.2).float()
    ld = (Ci + 3) // 4 * 4
    xb, xv = _padded(x, ld)
    if ld != Ci:
        xb[..., Ci:] = float("nan")
    old = _rand(tuple(ref.shape), 614, dev)
    y = old.clone()
    backend.lib.tune_conv_patch(0)            # (keep the patch / bank kernels out of the way: this is about the tiled kernel)
    backend.lib.tune_conv_x3_igemm(1)         # off by default (measured slower than exact fp32 on these latency-bound layers)
    try:
        with ops.precision_scope("mixed"):
            ops.conv2d_fwd(backend.lib, xv, w, b, ops.view(y), stride=s, dil=d, alpha=0.2, accumulate=True)
        name = backend.lib.last_kernel().decode()
        backend.sync()
    finally:
        backend.lib.tune_conv_patch(-1)
        backend.lib.tune_conv_x3_igemm(-1)
    if Co % 4 == 0:
        assert "conv_igemm_kernel" in name and "bf16x3" in name, name
    err = (y.cpu() - old.cpu() - ref).abs().max().item()
    assert err <= 4e-5 * max(1.0, ref.abs().max().item()), (err, name)


def test_wgrad_split_targets_follow_the_tuning_hook(backend):
    """mh_tune_wgrad_target_pct scales the pixel-split workgroup targets of the split counts resolved while it is set (DispNet's engine records its plans
    with 150); a plan stores the counts it was recorded with, and a launch with a stored count is valid whatever the hook says at that time."""
    import ctypes as C
    dev = backend.device
    H, W, Ci, Co = 40, 100, 96, 64                       # 4000 reduction pixels: the split count is target-limited, not pixel-limited
    x = _rand((1, H, W, Ci), 811, dev); gz = _rand((1, H, W, Co), 812, dev)
    xb, xv = _padded(x, Ci); zb, zv = _padded(gz, Co)
    d = ops.conv_desc(1, H, W, H, W, Ci, Co, 3, 3, 1, 1, 1, 1, 0, 0, Ci, Co, precision=1)
    counts = {}
    for pct in (0, 150, 50):
        backend.lib.tune_wgrad_target_pct(pct)
        sp = C.c_int32(0)
        backend.lib.conv2d_wgrad_partial(C.byref(d), ops._p(xv), ops._p(zv), zv.ld, None, C.byref(sp), None, None)
        counts[pct] = sp.value
    backend.lib.tune_wgrad_target_pct(0)
    assert counts[150] > counts[0] > counts[50] >= 1, counts
    # a count resolved under 150 % launches under the default setting
    # workspace (ABI 15): [splits][9*Ci*Co] filter partials, then [splits][Co] bias partials; db itself stays untouched when splits > 1
    ns, size = counts[150], 9 * Ci * Co
    ws = torch.full((ns * (size + Co),), float("nan"), device=dev); db = torch.zeros(Co, device=dev)
    sp = C.c_int32(ns)
    backend.lib.conv2d_wgrad_partial(C.byref(d), ops._p(xv), ops._p(zv), zv.ld, C.c_void_p(ws.data_ptr()), C.byref(sp), C.c_void_p(db.data_ptr()), None)
    backend.sync()
    xr = _bf(x.cpu()).double(); w0 = torch.zeros(3, 3, Ci, Co, dtype=torch.float64, requires_grad=True)
    (gw,) = torch.autograd.grad(T.conv2d(xr, w0, None, alpha=1.0), [w0], _bf(gz.cpu()).double())
    assert (ws[:ns * size].view(ns, size).cpu().double().sum(0).view(3, 3, Ci, Co) - gw).abs().max().item() <= 2e-5 * max(1.0, gw.abs().max().item())
    gb = gz.cpu().double().sum((0, 1, 2))
    assert ns > 1 and (db == 0).all() and (ws[ns * size:].view(ns, Co).cpu().double().sum(0) - gb).abs().max().item() <= 2e-5 * max(1.0, gb.abs().max().item())


def test_wgrad_reduce_small_gradient_many_splits(backend):
    """mh_wgrad_reduce on a small gradient with many splits (the 3-channel image layer: 432 values x 167 splits) takes the one-block path that shares
    the splits among the threads: same sums as the generic path (a 4096-value segment in the same launch), accumulate and overwrite forms."""
    dev = backend.device
    g = torch.Generator().manual_seed(5)
    segs, keep, refs = [], [], []
    for size, splits in ((432, 167), (4096, 3), (64, 40), (1024, 32)):
        ws = torch.randn(splits, size, generator=g).to(dev)
        dst = torch.full((size,), 2.0, device=dev)
        segs.append((ws.data_ptr(), dst.data_ptr(), size, splits)); keep += [ws, dst]
        refs.append((dst, ws.double().sum(0).cpu()))
    for acc in (False, True):
        ops.wgrad_reduce(backend.lib, segs, dev, keep, accumulate=acc)
        backend.sync()
        for dst, ref in refs:
            want = ref + (ref if acc else 0)           # second pass: dst (= ref after the first) + ref
            assert (dst.cpu().double() - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("case", [(1, 12, 20, 32, 1, False), (2, 9, 13, 32, 2, True), (1, 6, 20, 12, 1, True), (1, 24, 80, 64, 1, False)])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_conv_k1_dgrad_kernel(backend, case, mode):
    """conv_k1_dgrad_kernel: input gradient of a single-output-channel 3x3 conv (the disparity heads; mode 1 with K = 1), exact fp32 in every
    arithmetic mode, with the accumulating / leaky-mask epilogue and the bf16 shadow of dx for the streamed filter gradient."""
    B, H, W, Ci, dil, acc = case
    dev = backend.device
    w = _rand((3, 3, Ci, 1), 912, dev, 0.3)
    gz = _rand((B, H, W, 1), 914, dev)
    old = _rand((B, H, W, Ci), 915, dev)
    mref = _rand((B, H, W, Ci), 916, dev)
    ld = Ci + 4                                         # a concat neighbour behind the channels: must stay untouched
    dxb, dxv = _padded(old, ld)
    dxb[..., Ci:] = 7.0
    mb, mv = _padded(mref, ld)
    sh = ops.Shadow(B, H, W, Ci, dev)
    sh.t.fill_(3.0)
    ops.PRECISION_BWD = 1 if mode == "bf16" else None
    try:
        ops.conv2d_dgrad(backend.lib, ops.view(gz), w, dxv, dil=dil, accumulate=acc, mask_ref=mv, mask_alpha=0.2, shadow=sh)
    finally:
        ops.PRECISION_BWD = None
    name = backend.lib.last_kernel().decode()
    backend.sync()
    assert "conv_k1_dgrad_kernel" in name, name
    xr = torch.zeros(B, H, W, Ci, dtype=torch.float64, requires_grad=True)
    yr = T.conv2d(xr, w.cpu().double(), None, dilation=dil, alpha=1.0)
    (gx,) = torch.autograd.grad(yr, [xr], gz.cpu().double())
    exp = (((old.cpu().double() if acc else 0.0) + gx) * torch.where(mref.cpu() > 0, 1.0, 0.2)).float()
    sc = max(1.0, gx.abs().max().item())
    got = dxb[..., :Ci].cpu()
    assert (got - exp).abs().max().item() <= 3e-6 * sc
    assert (dxb[..., Ci:] == 7.0).all()
    assert torch.equal(sh.t[..., :Ci].cpu(), got.to(torch.bfloat16))             # RNE of exactly what went to dx
    assert (sh.t[..., Ci:].cpu().float() == 3.0).all()                              # the shadow's channel padding is the caster's business


ROWS_CASES = [   # (B, H, W, Cin, Cout): ragged strips (W % 32), ragged row blocks, one-row images, fewer channels than the MFMA tile
    (1, 12, 64, 16, 16), (2, 9, 45, 16, 16), (1, 1, 33, 16, 16), (1, 2, 31, 8, 16), (1, 17, 70, 16, 32), (1, 5, 32, 12, 24), (1, 40, 100, 16, 16),
]


@pytest.mark.parametrize("what", ["fwd-bf16", "fwd-x3", "dgrad-bf16"])
@pytest.mark.parametrize("case", ROWS_CASES)
def test_conv_rows_kernel(backend, case, what):
    """conv_rows_kernel (csrc/conv_rows.hip): the row-streaming kernel of the thin 3x3 stride-1 layers -- filter bank as the A operand in registers,
    pixels as the B operand straight from global memory, three rotating accumulators.  Same operand rounding as the other kernels of the precision
    code; compared with the fp64 oracle on identically rounded operands (bf16) / on the fp32 operands (split-bf16), and with the kernel the layer
    takes otherwise.  Epilogues: bias + leaky + shadow (forward), accumulate + leaky mask + shadow (input gradient), a concat neighbour behind the
    output channels."""
    B, H, W, Ci, Co = case
    dev = backend.device
    w = _rand((3, 3, Ci, Co), 712, dev, 0.2)
    b = _rand((Co,), 713, dev)
    prev = backend.lib.tune_conv_rows(1)
    try:
        if what.startswith("fwd"):
            x = _rand((B, H, W, Ci), 711, dev)
            ld = Ci + 4
            xb, xv = _padded(x, ld)
            xb[..., Ci:] = float("nan")
            x3 = what == "fwd-x3"
            yb = torch.full((B, H, W, Co + 4), 5.0, device=dev)
            yv = ops.View(yb, B, H, W, Co, Co + 4)
            y0 = torch.full((B, H, W, Co), float("nan"), device=dev)
            sh = ops.Shadow(B, H, W, Co, dev)
            with ops.precision_scope("mixed" if x3 else "bf16"):
                ops.conv2d_fwd(backend.lib, xv, w, b, yv, alpha=0.2, shadow=sh)
                name = backend.lib.last_kernel().decode()
                backend.lib.tune_conv_rows(0)
                ops.conv2d_fwd(backend.lib, xv, w, b, ops.view(y0), alpha=0.2)
                other = backend.lib.last_kernel().decode()
            backend.sync()
            assert "conv_rows_kernel<fwd,%s,s1>" % ("bf16x3" if x3 else "bf16") in name and "conv_rows" not in other, (name, other)
            if x3:
                ref = T.conv2d(x.cpu().double(), w.cpu().double(), b.cpu().double(), alpha=0.2).float()
                tol = 4e-5
            else:
                ref = T.conv2d(_bf(x.cpu()).double(), _bf(w.cpu()).double(), b.cpu().double(), alpha=0.2).float()
                tol = 2e-5
            sc = max(1.0, ref.abs().max().item())
            got = yb[..., :Co].cpu()
            assert (got - ref).abs().max().item() <= tol * sc
            if not x3:            # (without a rows kernel the split-bf16 request falls back to exact fp32 on these shapes)
                assert (got - y0.cpu()).abs().max().item() <= 2 * tol * sc
            assert (yb[..., Co:] == 5.0).all()
            assert torch.equal(sh.t[..., :Co].cpu(), got.to(torch.bfloat16))
        else:
            if Co > 16 or Ci % 8:
                pytest.skip("input gradient: the contraction runs over the OUTPUT channels of the forward layer (<= 16 for this kernel)")
            gz = _rand((B, H, W, Co), 714, dev)
            old = _rand((B, H, W, Ci), 715, dev)
            mref = _rand((B, H, W, Ci), 716, dev)
            ldx = Ci + 4
            zb, zv = _padded(gz, Co)
            mb, mv = _padded(mref, ldx)
            outs = []
            sh = ops.Shadow(B, H, W, Ci, dev)
            for rows in (True, False):
                backend.lib.tune_conv_rows(1 if rows else 0)
                dxb, dxv = _padded(old, ldx)
                dxb[..., Ci:] = 5.0
                ops.PRECISION_BWD = 1
                try:
                    ops.conv2d_dgrad(backend.lib, zv, w, dxv, accumulate=True, mask_ref=mv, mask_alpha=0.2, shadow=(sh if rows else None))
                finally:
                    ops.PRECISION_BWD = None
                name = backend.lib.last_kernel().decode()
                assert ("conv_rows_kernel<dgrad,bf16,s1>" in name) == rows, name
                backend.sync()
                assert (dxb[..., Ci:] == 5.0).all()
                outs.append(dxb[..., :Ci].cpu().clone())
            xr = torch.zeros(B, H, W, Ci, dtype=torch.float64, requires_grad=True)
            yr = T.conv2d(xr, _bf(w.cpu()).double(), None, alpha=1.0)
            (gx,) = torch.autograd.grad(yr, [xr], _bf(gz.cpu()).double())
            exp = ((old.cpu().double() + gx) * torch.where(mref.cpu() > 0, 1.0, 0.2)).float()
            sc = max(1.0, gx.abs().max().item())
            assert (outs[0] - exp).abs().max().item() <= 2e-5 * sc
            assert (outs[0] - outs[1]).abs().max().item() <= 4e-5 * sc
            assert torch.equal(sh.t[..., :Ci].cpu(), outs[0].to(torch.bfloat16))
            # the same launch with dz and the mask taken from their bf16 shadows (mh_conv2d_sh3): bit-identical, the fp32 dz is not read
            keep = []
            shz, shm = ops.Shadow(B, H, W, Co, dev), ops.Shadow(B, H, W, Ci, dev)
            ops.shadow_cast(backend.lib, [(zv, shz), (mv, shm)], dev, keep)
            backend.sync()
            zb.fill_(float("nan"))
            backend.lib.tune_conv_rows(1)
            dxb, dxv = _padded(old, ldx)
            ops.PRECISION_BWD = 1
            try:
                ops.conv2d_dgrad(backend.lib, zv, w, dxv, accumulate=True, mask_ref=mv, mask_alpha=0.2, dz_shadow=shz, mask_shadow=shm)
            finally:
                ops.PRECISION_BWD = None
            name = backend.lib.last_kernel().decode()
            backend.sync()
            assert "conv_rows_kernel<dgrad,bf16,s1,shadows>" in name, name
            assert torch.equal(dxb[..., :Ci].cpu(), outs[0])
    finally:
        backend.lib.tune_conv_rows(prev)


ROWS_S2_CASES = [   # (B, H, W, Cin, Cout): even sizes (pad 0 / 1), odd sizes (pad 1 / 1), the image layer (3 channels in a 4-float pixel)
    (1, 24, 64, 16, 32), (2, 18, 90, 16, 16), (1, 13, 35, 16, 32), (1, 20, 66, 3, 16), (2, 9, 33, 3, 16), (1, 2, 4, 8, 8),
]


@pytest.mark.parametrize("x3", [False, True], ids=["bf16", "x3"])
@pytest.mark.parametrize("case", ROWS_S2_CASES)
def test_conv_rows_kernel_stride2(backend, case, x3):
    """conv_rows_kernel<.., s2>: forward pass of the down-sampling layers (conv1 3 -> 16 on the padded image, conv3 16 -> 32): two new input rows per
    output row, the third stays in registers as the next row's first; TF 'SAME' padding for even and odd sizes; NaN in the padding channel of a
    3-channel pixel must not reach the MFMA."""
    B, H, W, Ci, Co = case
    dev = backend.device
    w = _rand((3, 3, Ci, Co), 722, dev, 0.2)
    b = _rand((Co,), 723, dev)
    x = _rand((B, H, W, Ci), 721, dev)
    ld = (Ci + 3) // 4 * 4 if Ci < 4 else Ci + 4
    xb, xv = _padded(x, ld)
    xb[..., Ci:] = float("nan")
    Ho, Wo, _, _ = ops.conv_geometry(H, W, 3, 3, 2, 1)
    y = torch.full((B, Ho, Wo, Co), float("nan"), device=dev); y0 = torch.full((B, Ho, Wo, Co), float("nan"), device=dev)
    sh = ops.Shadow(B, Ho, Wo, Co, dev)
    prev = backend.lib.tune_conv_rows(1)
    try:
        if Ci < 4:
            xb[..., Ci:] = 0.0                      # (the other kernels multiply the padding channel by a zero weight: keep it finite for them)
        with ops.precision_scope("mixed" if x3 else "bf16"):
            backend.lib.tune_conv_rows(0)
            ops.conv2d_fwd(backend.lib, xv, w, b, ops.view(y0), stride=2, alpha=0.2)
            other = backend.lib.last_kernel().decode()
            backend.lib.tune_conv_rows(1)
            if Ci < 4:
                xb[..., Ci:] = float("nan")
            ops.conv2d_fwd(backend.lib, xv, w, b, ops.view(y), stride=2, alpha=0.2, shadow=sh)
            name = backend.lib.last_kernel().decode()
        backend.sync()
    finally:
        backend.lib.tune_conv_rows(prev)
    assert "conv_rows_kernel<fwd,%s,s2>" % ("bf16x3" if x3 else "bf16") in name and "conv_rows" not in other, (name, other)
    if x3:
        ref = T.conv2d(x.cpu().double(), w.cpu().double(), b.cpu().double(), stride=2, alpha=0.2).float()
        tol = 4e-5
    else:
        ref = T.conv2d(_bf(x.cpu()).double(), _bf(w.cpu()).double(), b.cpu().double(), stride=2, alpha=0.2).float()
        tol = 2e-5
    sc = max(1.0, ref.abs().max().item())
    assert (y.cpu() - ref).abs().max().item() <= tol * sc
    if not x3:
        assert (y.cpu() - y0.cpu()).abs().max().item() <= 2 * tol * sc
    assert torch.equal(sh.t[..., :Co].cpu(), y.cpu().to(torch.bfloat16))


@pytest.mark.parametrize("case", [(1, 12, 20, 128, 128, 1), (1, 9, 21, 64, 96, 2), (2, 10, 18, 128, 40, 1), (1, 14, 19, 96, 64, 4), (1, 9, 33, 64, 32, 1),
                                  (1, 12, 20, 38, 128, 1), (2, 9, 17, 70, 128, 1)])     # 38 / 70 gradient channels in rows of 40 / 72 (the estimators' first layers)
def test_conv_patch_input_gradient_from_the_shadow(backend, case):
    """mh_conv2d_sh2: the patch-staged input-gradient kernel stages the bf16 shadow of dz (written by an earlier epilogue / mh_shadow_cast) instead of
    converting the fp32 tensor -- the same round-to-nearest-even done earlier, so dx is BIT-identical; the fp32 dz is not read at all (poisoned here)."""
    B, H, W, Ci, Co, dil = case
    dev = backend.device
    w = _rand((3, 3, Ci, Co), 812, dev, 0.2)
    gz = _rand((B, H, W, Co), 814, dev)
    old = _rand((B, H, W, Ci), 815, dev)
    mref = _rand((B, H, W, Ci), 816, dev)
    ldx = (Ci + 3) // 4 * 4
    zb, zv = _padded(gz, Co)
    mb, mv = _padded(mref, ldx)
    keep = []
    shz = ops.Shadow(B, H, W, Co, dev)
    ops.shadow_cast(backend.lib, [(zv, shz)], dev, keep)
    backend.sync()
    prev = backend.lib.tune_conv_patch(128)            # force the patch kernel at these small sizes
    outs = []
    try:
        for use_shadow in (False, True):
            dxb, dxv = _padded(old, ldx)
            if use_shadow:
                zb.fill_(float("nan"))
            ops.PRECISION_BWD = 1
            try:
                ops.conv2d_dgrad(backend.lib, zv, w, dxv, dil=dil, accumulate=True, mask_ref=mv, mask_alpha=0.2, dz_shadow=(shz if use_shadow else None))
            finally:
                ops.PRECISION_BWD = None
            name = backend.lib.last_kernel().decode()
            assert "conv_patch_kernel" in name and "dgrad" in name, name
            backend.sync()
            outs.append(dxb[..., :Ci].cpu().clone())
    finally:
        backend.lib.tune_conv_patch(-1)
    assert torch.isfinite(outs[1]).all() and torch.equal(outs[0], outs[1])
    xr = torch.zeros(B, H, W, Ci, dtype=torch.float64, requires_grad=True)
    yr = T.conv2d(xr, _bf(w.cpu()).double(), None, dilation=dil, alpha=1.0)
    (gx,) = torch.autograd.grad(yr, [xr], _bf(gz.cpu()).double())
    exp = ((old.cpu().double() + gx) * torch.where(mref.cpu() > 0, 1.0, 0.2)).float()
    assert (outs[1] - exp).abs().max().item() <= 2e-5 * max(1.0, gx.abs().max().item())


@pytest.mark.parametrize("case", [(1, 12, 20, 32), (2, 9, 13, 32), (1, 24, 80, 64)])
def test_conv2d_head_extra_destinations(backend, case):
    """mh_conv2d_head: the single-output-channel forward conv of a disparity head also stores its result in a channel slot of a wider buffer and in
    a second map, bit-identical to mh_conv2d; neighbours of the slot stay untouched."""
    B, H, W, Ci = case
    dev = backend.device
    x = _rand((B, H, W, Ci), 511, dev)
    w = _rand((3, 3, Ci, 1), 512, dev, 0.3)
    b = _rand((1,), 513, dev)
    y0 = torch.full((B, H, W, 1), float("nan"), device=dev)
    ops.conv2d_fwd(backend.lib, ops.view(x), w, b, ops.view(y0))
    y = torch.full((B, H, W, 1), float("nan"), device=dev)
    cat = torch.full((B, H, W, 36), 3.0, device=dev)            # e.g. [features (33) | disparity | padding]
    acc = torch.full((B, H, W, 1), float("nan"), device=dev)
    slot = ops.View(cat, B, H, W, 36, 36).slice(33, 34)
    ops.conv2d_head(backend.lib, ops.view(x), w, b, ops.view(y), copies=(slot, ops.view(acc)))
    assert "conv_n1_fwd_kernel" in backend.lib.last_kernel().decode()
    backend.sync()
    assert torch.equal(y.cpu(), y0.cpu()) and torch.equal(acc.cpu(), y0.cpu()) and torch.equal(cat[..., 33:34].cpu(), y0.cpu())
    assert (cat[..., :33] == 3.0).all() and (cat[..., 34:] == 3.0).all()
