"""mh_conv2d_planes / mh_plane_split / mh_pack_weights(trans = 2) (csrc/conv_planes.hip): the split-bf16 forward pass of the stride-1 3x3 layers
(Nets/sharedLayers.py:54-77 for the layers of Nets/MadNet.py:73-171) from PRE-SPLIT operands -- activations as hi / lo bf16 planes staged by LDS DMA,
weights from a fragment bank in the 32x32x16 MFMA register image.  Checked against
  * the fp64 oracle at the 2^-16 level (the arithmetic claim of the split-bf16 forward mode),
  * the existing split-bf16 kernel on the same fp32 inputs (same three products per element, another summation order: fp32 round-off apart),
  * itself: output planes == the split of the fp32 result, bit for bit; untouched padding channels stay zero."""
import pytest
import torch

from madnet_hip import ops
from oracle import tf_ops as T


def _rand(shape, seed, dev, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def _split_ref(x):
    """hi = bf16(x), lo = bf16(x - hi) as the kernels compute them (round to nearest even)"""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi, lo


def _planes_of(lib, x, dev, keep):
    """Planes of an fp32 NHWC tensor through mh_plane_split (+ the check of that kernel against torch's bf16 rounding)"""
    B, H, W, C = x.shape
    pl = ops.Planes(ops.Shadow(B, H, W, C, dev), dev)
    ops.plane_split(lib, [(ops.view(x), pl)], dev, keep)
    hi, lo = _split_ref(x.cpu())
    assert torch.equal(pl.hi.t.cpu()[..., :C], hi) and torch.equal(pl.lo.t.cpu()[..., :C], lo)
    assert not pl.hi.t.cpu()[..., C:].float().abs().sum() and not pl.lo.t.cpu()[..., C:].float().abs().sum()
    return pl


# (B, H, W, Cin, Cout, dil, variant): variant = mh_tune_conv_planes tile variant (0 = heuristic; 1..4 see csrc/conv_planes.hip dispatch_planes_k)
CASES = [
    (1, 9, 37, 128, 128, 1, 1), (1, 9, 37, 128, 128, 1, 2), (1, 11, 21, 128, 128, 1, 3), (1, 11, 21, 128, 128, 1, 4),
    (1, 12, 40, 128, 128, 2, 0), (2, 7, 33, 128, 96, 1, 1), (1, 20, 18, 128, 96, 1, 3), (1, 10, 35, 96, 64, 1, 1), (1, 18, 20, 96, 64, 4, 2),
    (1, 9, 34, 64, 32, 1, 1), (1, 19, 17, 64, 32, 1, 2), (2, 8, 40, 32, 32, 1, 0),
    (1, 8, 33, 38, 128, 1, 1),       # the estimators' first layer: 38 channels in rows of 64 halfs, three 16-channel steps per tap
    (1, 8, 33, 33, 128, 1, 3), (1, 9, 20, 70, 128, 1, 1), (1, 6, 20, 64, 64, 16, 1), (1, 12, 40, 128, 96, 8, 0),
    # + 16: the staggered form of the 128 x 128 tile (two out-of-phase wave groups, wave-local epilogue: conv_planes_kernel_stg)
    (1, 9, 37, 128, 128, 1, 17), (1, 11, 21, 128, 128, 2, 19), (1, 8, 33, 38, 128, 1, 17), (2, 9, 40, 33, 128, 1, 19), (2, 7, 33, 128, 96, 1, 17), (1, 10, 35, 96, 64, 1, 19),
]


@pytest.mark.parametrize("case", CASES)
def test_conv2d_planes_vs_oracle_and_split_bf16_kernel(backend, case):
    B, H, W, Ci, Co, dil, variant = case
    lib, dev = backend.lib, backend.device
    x = _rand((B, H, W, Ci), 211, dev)
    w = _rand((3, 3, Ci, Co), 212, dev, 0.2)
    b = _rand((Co,), 213, dev)
    y_ref = T.conv2d(x.cpu().double(), w.cpu().double(), b.cpu().double(), stride=1, dilation=dil, alpha=0.2).float()
    keep = []
    assert ops.conv2d_planes_ok(lib, ops.view(x), w, dil)
    xp = _planes_of(lib, x, dev, keep)
    bank = torch.full((ops.pack_bytes(w, 2, 2) // 4,), float("nan"), device=dev)
    assert ops.pack_bytes(w, 2, 2) == lib.pack32_bytes(9, Ci, Co)
    ops.pack_weights(lib, [(w, bank, 2, 2)], dev, keep)
    y = torch.full((B, H, W, Co), float("nan"), device=dev)
    yp = ops.Planes(ops.Shadow(B, H, W, Co, dev), dev)
    lib.tune_conv_planes(variant)
    try:
        ops.conv2d_planes(lib, xp, w, bank, b, out=ops.view(y), out_planes=yp, dil=dil, alpha=0.2)
        name = lib.last_kernel().decode()
        backend.sync()
        # planes only (no fp32 result) and fp32 only: the same values
        yp2 = ops.Planes(ops.Shadow(B, H, W, Co, dev), dev)
        ops.conv2d_planes(lib, xp, w, bank, b, out=None, out_planes=yp2, dil=dil, alpha=0.2)
        y3 = torch.full((B, H, W, Co), float("nan"), device=dev)
        ops.conv2d_planes(lib, xp, w, bank, b, out=ops.view(y3), out_planes=None, dil=dil, alpha=0.2)
        backend.sync()
    finally:
        assert lib.tune_conv_planes(0) == 3
    assert "conv_planes_kernel" in name and (("staggered" in name) == bool(variant & 16)), name
    yc = y.cpu()
    assert torch.isfinite(yc).all()
    scale = max(1.0, y_ref.abs().max().item())
    err = (yc - y_ref).abs().max().item()
    assert err <= 4e-5 * scale, (err, name)
    hi, lo = _split_ref(yc)
    assert torch.equal(yp.hi.t.cpu()[..., :Co], hi) and torch.equal(yp.lo.t.cpu()[..., :Co], lo)
    assert torch.equal(yp2.hi.t.cpu(), yp.hi.t.cpu()) and torch.equal(yp2.lo.t.cpu(), yp.lo.t.cpu()) and torch.equal(y3.cpu(), yc)
    # the existing split-bf16 forward kernel on the fp32 tensor: identical products, another summation order
    if Ci % 4 == 0:
        y0 = torch.full((B, H, W, Co), float("nan"), device=dev)
        lib.tune_conv_patch(128)
        try:
            with ops.precision_scope("mixed"):
                ops.conv2d_fwd(lib, ops.view(x), w, b, ops.view(y0), stride=1, dil=dil, alpha=0.2)
            backend.sync()
        finally:
            lib.tune_conv_patch(-1)
        d0 = (yc - y0.cpu()).abs().max().item()
        assert d0 <= 4e-5 * scale, d0


def test_conv2d_planes_chain_equals_fp32_reference_chain(backend):
    """two layers back to back through planes only (no fp32 tensor in between): what the engines' forward chains do"""
    lib, dev = backend.lib, backend.device
    B, H, W = 1, 10, 36
    x = _rand((B, H, W, 64), 311, dev)
    w1, b1 = _rand((3, 3, 64, 96), 312, dev, 0.1), _rand((96,), 313, dev)
    w2, b2 = _rand((3, 3, 96, 64), 314, dev, 0.1), _rand((64,), 315, dev)
    keep = []
    xp = _planes_of(lib, x, dev, keep)
    banks = [torch.zeros(ops.pack_bytes(w, 2, 2) // 4, device=dev) for w in (w1, w2)]
    ops.pack_weights(lib, [(w1, banks[0], 2, 2), (w2, banks[1], 2, 2)], dev, keep)
    mid = ops.Planes(ops.Shadow(B, H, W, 96, dev), dev)
    y = torch.zeros(B, H, W, 64, device=dev)
    ops.conv2d_planes(lib, xp, w1, banks[0], b1, out=None, out_planes=mid, dil=1, alpha=0.2)
    ops.conv2d_planes(lib, mid, w2, banks[1], b2, out=ops.view(y), out_planes=None, dil=2, alpha=0.2)
    backend.sync()
    r1 = T.conv2d(x.cpu().double(), w1.cpu().double(), b1.cpu().double(), stride=1, dilation=1, alpha=0.2)
    r2 = T.conv2d(r1, w2.cpu().double(), b2.cpu().double(), stride=1, dilation=2, alpha=0.2).float()
    err = (y.cpu() - r2).abs().max().item()
    assert err <= 1e-4 * max(1.0, r2.abs().max().item()), err


def test_conv2d_planes_argument_checks(backend):
    from madnet_hip import _ffi
    import ctypes as C
    lib, dev = backend.lib, backend.device
    x = torch.zeros(1, 4, 8, 64, device=dev)
    w = torch.zeros(3, 3, 64, 32, device=dev)
    keep = []
    xp = _planes_of(lib, x, dev, keep)
    bank = torch.zeros(ops.pack_bytes(w, 2, 2) // 4, device=dev)
    y = torch.zeros(1, 4, 8, 32, device=dev)
    with pytest.raises(_ffi.MadnetHipError):          # no output at all
        ops.conv2d_planes(lib, xp, w, bank, None, out=None, out_planes=None)
    d = ops.conv_desc(1, 4, 8, 4, 8, 64, 32, 3, 3, 2, 1, 1, 1, 0, 0, 0, 32, precision=2)     # stride 2: no instance
    assert lib.conv2d_planes_ok(C.byref(d)) == 0
    rc = lib._raw_mh_conv2d_planes(C.byref(d), xp.hi.ptr, xp.lo.ptr, xp.ld, bank.data_ptr(), None, y.data_ptr(), None, None, 0, None)
    assert rc == -3 and b"mh_conv2d_planes" in lib.last_error()
    d = ops.conv_desc(1, 4, 8, 4, 8, 64, 32, 3, 3, 1, 1, 1, 1, 0, 0, 0, 32, precision=2)
    assert lib.conv2d_planes_ok(C.byref(d)) == 1
    rc = lib._raw_mh_conv2d_planes(C.byref(d), xp.hi.ptr, xp.lo.ptr, 40, bank.data_ptr(), None, y.data_ptr(), None, None, 0, None)   # plane stride < K
    assert rc == -1
    rc = lib._raw_mh_conv2d_planes(C.byref(d), xp.hi.ptr + 2, xp.lo.ptr, xp.ld, bank.data_ptr(), None, y.data_ptr(), None, None, 0, None)
    assert rc == -2
    assert not ops.conv2d_planes_ok(lib, ops.view(torch.zeros(1, 4, 8, 200, device=dev)), torch.zeros(3, 3, 200, 32), 1)      # K16 = 13 has 64 columns only
    assert not ops.conv2d_planes_ok(lib, ops.view(x), torch.zeros(3, 3, 64, 20), 1)                                             # N not a multiple of 8
    assert ops.conv2d_planes_ok(lib, ops.view(x), torch.zeros(3, 3, 64, 160), 1)                                                # two 128-column tiles


# (B, H, W, Cin, Cout, dil, variant): the input gradients of the layers above (reduction over Cout, columns = Cin)
BWD_CASES = [
    (1, 9, 37, 128, 128, 1, 0), (1, 11, 21, 128, 128, 2, 3), (1, 12, 40, 128, 96, 1, 2), (2, 7, 33, 96, 64, 1, 0), (1, 18, 20, 96, 64, 4, 4),
    (1, 9, 34, 64, 32, 1, 0), (2, 8, 40, 32, 32, 1, 1), (1, 10, 35, 64, 64, 1, 5), (1, 6, 20, 64, 64, 16, 0),
    (1, 9, 37, 128, 128, 1, 33), (1, 11, 21, 128, 128, 2, 35), (2, 6, 40, 128, 96, 1, 33),       # + 32: staggered wave groups
]


@pytest.mark.parametrize("case", BWD_CASES)
def test_conv2d_planes_bwd_vs_oracle_and_bf16_input_gradient(backend, case):
    """mh_conv2d_planes_bwd: dx = conv2d_backprop_input(dz, w) * leaky'(x) from the bf16 shadow of dz and the one-plane mirrored / transposed bank
    (mh_pack_weights trans = 3).  Against the oracle run on identically rounded operands (products of bf16 values are exact in fp32: only the fp32
    summation order differs) and against the existing bf16 input-gradient kernel on the fp32 tensors; the shadow of dx is bf16(dx) bit for bit."""
    B, H, W, Ci, Co, dil, variant = case
    lib, dev = backend.lib, backend.device
    dz = _rand((B, H, W, Co), 411, dev)
    w = _rand((3, 3, Ci, Co), 412, dev, 0.2)
    x = _rand((B, H, W, Ci), 413, dev)           # the forward layer's input: only its sign matters (leaky mask)
    keep = []
    # oracle: autograd of the forward conv on bf16-rounded dz / w (what precision code 1 computes)
    wq = w.cpu().to(torch.bfloat16).double()
    dzq = dz.cpu().to(torch.bfloat16).double()
    xin = torch.zeros(B, H, W, Ci, dtype=torch.float64, requires_grad=True)
    y = T.conv2d(xin, wq, None, stride=1, dilation=dil, alpha=1.0)
    (g_ref,) = torch.autograd.grad(y, xin, dzq)
    g_ref = (g_ref * torch.where(x.cpu().double() > 0, 1.0, 0.2)).float()
    assert ops.conv2d_planes_bwd_ok(lib, ops.view(x), w, dil)
    dzs = ops.Shadow(B, H, W, Co, dev); xs = ops.Shadow(B, H, W, Ci, dev)
    ops.shadow_cast(lib, [(ops.view(dz), dzs), (ops.view(x), xs)], dev, keep)
    bank = torch.full((ops.pack_bytes(w, 1, 3) // 4,), float("nan"), device=dev)
    ops.pack_weights(lib, [(w, bank, 1, 3)], dev, keep)
    dx = torch.full((B, H, W, Ci), float("nan"), device=dev)
    dxs = ops.Shadow(B, H, W, Ci, dev)
    lib.tune_conv_planes(variant)
    try:
        ops.conv2d_planes_bwd(lib, dzs, w, bank, dx=ops.view(dx), dx_shadow=dxs, mask_shadow=xs, mask_alpha=0.2, dil=dil)
        name = lib.last_kernel().decode()
        dxs2 = ops.Shadow(B, H, W, Ci, dev)
        ops.conv2d_planes_bwd(lib, dzs, w, bank, dx=None, dx_shadow=dxs2, mask_shadow=xs, mask_alpha=0.2, dil=dil)       # shadow only
        backend.sync()
    finally:
        assert lib.tune_conv_planes(0) == 2
    assert "conv_planes_kernel" in name and ",bf16>" in name and (("staggered" in name) == bool(variant & 32)), name
    dxc = dx.cpu()
    scale = max(1.0, g_ref.abs().max().item())
    assert torch.isfinite(dxc).all() and (dxc - g_ref).abs().max().item() <= 2e-5 * scale, ((dxc - g_ref).abs().max().item(), name)
    assert torch.equal(dxs.t.cpu()[..., :Ci], dxc.to(torch.bfloat16)) and torch.equal(dxs2.t.cpu(), dxs.t.cpu())
    # the existing bf16 input-gradient path on the fp32 tensors (same rounding of both operands)
    dx0 = torch.zeros(B, H, W, Ci, device=dev)
    with ops.precision_scope("bf16"):
        ops.conv2d_dgrad(lib, ops.view(dz), w, ops.view(dx0), dil=dil, mask_ref=ops.view(x), mask_alpha=0.2)
    backend.sync()
    assert (dxc - dx0.cpu()).abs().max().item() <= 2e-5 * scale


# ---- DispNet's layers (Nets/DispNet.py:75-152): K-chunked reductions (> 128 channels), one-plane (plain bf16) forward, wide gradients ----------------
# (B, H, W, Cin, Cout): forward, plain bf16 from the hi plane and the one-plane chunk-major bank
CK_FWD_CASES = [(1, 6, 33, 256, 256), (1, 5, 20, 385, 128), (1, 3, 35, 512, 256), (2, 4, 9, 1025, 128), (1, 7, 40, 136, 64)]


@pytest.mark.parametrize("case", CK_FWD_CASES)
def test_conv2d_planes_chunked_bf16_forward(backend, case):
    """mh_conv2d_planes(precision 1) on reductions over more than 128 channels: conv_planes_ck_kernel (a loader wave streams 64-channel patch chunks
    through three LDS buffers).  Against the fp64 oracle on bf16-rounded operands (only the fp32 summation order differs); hi-plane output ==
    bf16(out) bit for bit."""
    B, H, W, Ci, Co = case
    lib, dev = backend.lib, backend.device
    x = _rand((B, H, W, Ci), 511, dev)
    w = _rand((3, 3, Ci, Co), 512, dev, 0.05)
    bias = _rand((Co,), 513, dev, 0.1)
    keep = []
    ref = T.conv2d(x.cpu().to(torch.bfloat16).double(), w.cpu().to(torch.bfloat16).double(), bias.cpu().double(), stride=1, dilation=1, alpha=0.1).float()
    assert ops.planes_kc16(Ci) == 4 == lib.planes_kc16(Ci)
    assert ops.conv2d_planes_ok(lib, ops.view(x), w, 1, bf16=True)
    xs = ops.Shadow(B, H, W, Ci, dev)
    ops.shadow_cast(lib, [(ops.view(x), xs)], dev, keep)
    bank = torch.full((ops.pack_bytes(w, 1, 2) // 4,), float("nan"), device=dev)
    ops.pack_weights(lib, [(w, bank, 1, 2)], dev, keep)
    out = torch.full((B, H, W, Co), float("nan"), device=dev)
    os_ = ops.Shadow(B, H, W, Co, dev)
    lib.tune_conv_planes(0)
    ops.conv2d_planes(lib, xs, w, bank, bias, out=ops.view(out), out_planes=os_, alpha=0.1, bf16=True)
    name = lib.last_kernel().decode()
    backend.sync()
    assert lib.tune_conv_planes(0) == 1
    assert "conv_planes_ck_kernel" in name and ",bf16," in name, name
    oc = out.cpu()
    scale = max(1.0, ref.abs().max().item())
    assert torch.isfinite(oc).all() and (oc - ref).abs().max().item() <= 3e-5 * scale, ((oc - ref).abs().max().item(), name)
    assert torch.equal(os_.t.cpu()[..., :Co], oc.to(torch.bfloat16))


# (B, H, W, Cin, Cout, mask range or None): input gradients -- reduction over Cout (chunked beyond 128), Cin columns in 128-column tiles, rows of
# Cin rounded up to 8 floats; the leaky mask of ONE member of a concat (DispNet's up-sampling blocks: [skip | deconv | up_predict])
CK_BWD_CASES = [(1, 6, 33, 256, 256, None), (1, 5, 20, 385, 128, (256, 384)), (1, 3, 17, 1025, 512, (512, 1024)), (1, 7, 36, 193, 64, (0, 128)),
                (1, 9, 40, 97, 32, (64, 96)), (2, 4, 9, 512, 512, None)]


@pytest.mark.parametrize("case", CK_BWD_CASES)
def test_conv2d_planes_bwd_wide_and_chunked(backend, case):
    B, H, W, Ci, Co, mrange = case
    lib, dev = backend.lib, backend.device
    dz = _rand((B, H, W, Co), 611, dev)
    w = _rand((3, 3, Ci, Co), 612, dev, 0.05)
    x = _rand((B, H, W, Ci), 613, dev)
    keep = []
    wq = w.cpu().to(torch.bfloat16).double()
    dzq = dz.cpu().to(torch.bfloat16).double()
    xin = torch.zeros(B, H, W, Ci, dtype=torch.float64, requires_grad=True)
    y = T.conv2d(xin, wq, None, stride=1, dilation=1, alpha=1.0)
    (g_ref,) = torch.autograd.grad(y, xin, dzq)
    mk = torch.where(x.cpu().double() > 0, 1.0, 0.1)
    if mrange is not None:
        keepc = torch.ones(Ci, dtype=torch.bool); keepc[mrange[0]:mrange[1]] = False
        mk[..., keepc] = 1.0
    g_ref = (g_ref * mk).float()
    dzs = ops.Shadow(B, H, W, Co, dev); xs = ops.Shadow(B, H, W, Ci, dev)
    ops.shadow_cast(lib, [(ops.view(dz), dzs), (ops.view(x), xs)], dev, keep)
    bank = torch.full((ops.pack_bytes(w, 1, 3) // 4,), float("nan"), device=dev)
    ops.pack_weights(lib, [(w, bank, 1, 3)], dev, keep)
    ld = (Ci + 7) // 8 * 8
    dxb = torch.full((B, H, W, ld), float("nan"), device=dev)
    dx = ops.View(dxb, B, H, W, Ci, ld)
    assert ops.conv2d_planes_bwd_ok(lib, dx, w, 1)
    if Ci % 8:          # rows shorter than Cin rounded up to 8 floats cannot take the 8-column stores
        assert not ops.conv2d_planes_bwd_ok(lib, ops.view(x), w, 1)
    lib.tune_conv_planes(0)
    ops.conv2d_planes_bwd(lib, dzs, w, bank, dx=dx, mask_shadow=xs, mask_alpha=0.1, mask_range=(mrange or (0, 0)))
    name = lib.last_kernel().decode()
    backend.sync()
    assert lib.tune_conv_planes(0) == 1
    assert ("conv_planes_ck_kernel" in name) == (Co > 128), name
    dxc = dxb.cpu()[..., :Ci]
    scale = max(1.0, g_ref.abs().max().item())
    assert torch.isfinite(dxc).all() and (dxc - g_ref).abs().max().item() <= 3e-5 * scale, ((dxc - g_ref).abs().max().item(), name)
    assert not dxb.cpu()[..., Ci:].abs().sum()                     # the row padding is written as zeros


@pytest.mark.parametrize("case", [(1, 6, 40, 193, 64), (1, 9, 37, 97, 32)])
def test_conv2d_planes_split_bf16_dispnet_iconv_shapes(backend, case):
    """the whole-K split-bf16 instances of DispNet's two finest iconv layers (193 -> 64: K16 13, 97 -> 32: K16 7) at the 2^-16 level"""
    B, H, W, Ci, Co = case
    lib, dev = backend.lib, backend.device
    x = _rand((B, H, W, Ci), 711, dev)
    w = _rand((3, 3, Ci, Co), 712, dev, 0.05)
    bias = _rand((Co,), 713, dev, 0.1)
    keep = []
    ref = T.conv2d(x.cpu().double(), w.cpu().double(), bias.cpu().double(), stride=1, dilation=1, alpha=1.0).float()
    assert ops.planes_kc16(Ci) == 0 == lib.planes_kc16(Ci)
    assert ops.conv2d_planes_ok(lib, ops.view(x), w, 1)
    xp = _planes_of(lib, x, dev, keep)
    bank = torch.full((ops.pack_bytes(w, 2, 2) // 4,), float("nan"), device=dev)
    ops.pack_weights(lib, [(w, bank, 2, 2)], dev, keep)
    out = torch.full((B, H, W, Co), float("nan"), device=dev)
    ops.conv2d_planes(lib, xp, w, bank, bias, out=ops.view(out), alpha=1.0)
    name = lib.last_kernel().decode()
    backend.sync()
    lib.tune_conv_planes(0)
    oc = out.cpu()
    scale = max(1.0, ref.abs().max().item())
    assert "conv_planes_kernel" in name and "bf16x3" in name, name
    assert torch.isfinite(oc).all() and (oc - ref).abs().max().item() <= 6e-5 * scale, ((oc - ref).abs().max().item(), name)


# (B, Hz, Wz, Cin, Cout): input gradient of the STRIDE-2 3x3 layers (MADNet pyramid conv3 16 -> 32, conv5 32 -> 64; dz is Hz x Wz, dx 2Hz x 2Wz)
# ... and (round 6, last field k = 5) of DispNet's 5x5 stride-2 conv2 (64 -> 128; 'SAME' pads 1 in front: the patch has a row / column on both sides)
S2_BWD_CASES = [(1, 5, 33, 16, 32), (2, 8, 40, 32, 64), (1, 3, 70, 24, 32), (1, 24, 80, 16, 32), (1, 5, 33, 64, 128, 5), (2, 8, 40, 64, 128, 5), (1, 3, 70, 48, 128, 5),
                (1, 5, 40, 145, 256, 5), (1, 4, 33, 72, 256, 5)]            # DispNet conv3: 145 (= 64 + 81) gradient columns, reduction over 256 channels


@pytest.mark.parametrize("case", S2_BWD_CASES)
def test_conv2d_planes_bwd_stride2(backend, case):
    """mh_conv2d_planes_bwd on a stride-2 layer (conv_planes_s2bwd_kernel: the four parity classes of the output pixels as four small convolutions over one
    dz patch): against the autograd of the oracle's stride-2 conv on bf16-rounded operands (fp32 summation order apart) and the tiled bf16 input-gradient
    kernel; the shadow of dx is bf16(dx) bit for bit."""
    B, Hz, Wz, Ci, Co = case[:5]
    k = case[5] if len(case) > 5 else 3
    H, W = 2 * Hz, 2 * Wz
    lib, dev = backend.lib, backend.device
    dz = _rand((B, Hz, Wz, Co), 811, dev)
    w = _rand((k, k, Ci, Co), 812, dev, 0.2)
    x = _rand((B, H, W, Ci), 813, dev)
    keep = []
    wq = w.cpu().to(torch.bfloat16).double()
    dzq = dz.cpu().to(torch.bfloat16).double()
    xin = torch.zeros(B, H, W, Ci, dtype=torch.float64, requires_grad=True)
    y = T.conv2d(xin, wq, None, stride=2, dilation=1, alpha=1.0)
    assert tuple(y.shape) == (B, Hz, Wz, Co)
    (g_ref,) = torch.autograd.grad(y, xin, dzq)
    g_ref = (g_ref * torch.where(x.cpu().double() > 0, 1.0, 0.2)).float()
    ld = (Ci + 7) // 8 * 8
    dxb = torch.full((B, H, W, ld), float("nan"), device=dev)
    dx = ops.View(dxb, B, H, W, Ci, ld)
    assert ops.conv2d_planes_bwd_ok(lib, dx, w, 1, stride=2)
    dzs = ops.Shadow(B, Hz, Wz, Co, dev); xs = ops.Shadow(B, H, W, Ci, dev); dxs = ops.Shadow(B, H, W, Ci, dev)
    ops.shadow_cast(lib, [(ops.view(dz), dzs), (ops.view(x), xs)], dev, keep)
    bank = torch.full((ops.pack_bytes(w, 1, 3) // 4,), float("nan"), device=dev)
    ops.pack_weights(lib, [(w, bank, 1, 3)], dev, keep)
    lib.tune_conv_planes(0)
    ops.conv2d_planes_bwd(lib, dzs, w, bank, dx=dx, dx_shadow=dxs, mask_shadow=xs, mask_alpha=0.2, stride=2)
    name = lib.last_kernel().decode()
    backend.sync()
    assert lib.tune_conv_planes(0) == 1 and "conv_planes_s2bwd_kernel" in name, name
    dxc = dxb.cpu()[..., :Ci]
    scale = max(1.0, g_ref.abs().max().item())
    assert torch.isfinite(dxc).all() and (dxc - g_ref).abs().max().item() <= 2e-5 * scale, ((dxc - g_ref).abs().max().item(), name)
    assert torch.equal(dxs.t.cpu()[..., :Ci], dxc.to(torch.bfloat16))
    dx0 = torch.zeros(B, H, W, Ci, device=dev)
    with ops.precision_scope("bf16"):
        ops.conv2d_dgrad(lib, ops.view(dz), w, ops.view(dx0), stride=2, mask_ref=ops.view(x), mask_alpha=0.2)
    backend.sync()
    assert (dxc - dx0.cpu()).abs().max().item() <= 2e-5 * scale
    if True:
        # accumulating form (round 6: DispNet's conv1a is a skip connection, MADNet's conv5 reads a cost-volume level -- the gradient map already holds a contribution): (old + new) * mask
        old = _rand((B, H, W, Ci), 814, dev)
        dxb2 = torch.zeros(B, H, W, ld, device=dev); dxb2[..., :Ci] = old
        ops.conv2d_planes_bwd(lib, dzs, w, bank, dx=ops.View(dxb2, B, H, W, Ci, ld), mask_shadow=xs, mask_alpha=0.2, stride=2, accumulate=True)
        backend.sync()
        xr = torch.zeros(B, H, W, Ci, dtype=torch.float64, requires_grad=True)
        (g_raw,) = torch.autograd.grad(T.conv2d(xr, wq, None, stride=2, dilation=1, alpha=1.0), xr, dzq)
        exp = ((old.cpu().double() + g_raw) * torch.where(x.cpu().double() > 0, 1.0, 0.2)).float()
        assert (dxb2.cpu()[..., :Ci] - exp).abs().max().item() <= 2e-5 * max(1.0, exp.abs().max().item())


# (B, H, W, Cin, Cout, k): H, W even; the shapes with an instance -- DispNet conv2 (5x5, 64 -> 128) and the 3x3 down-sampling layers (16 -> 32, 32 -> 64, 64 -> 96)
S2_CASES = [(1, 12, 72, 64, 128, 5), (2, 6, 68, 64, 128, 5), (1, 18, 40, 64, 128, 5), (1, 16, 66, 16, 32, 3), (1, 10, 132, 32, 64, 3), (2, 6, 68, 64, 96, 3),
            (1, 8, 72, 145, 256, 5)]                     # DispNet conv3 (the one-plane instance only: it runs plain bf16 in every mode that uses planes)


@pytest.mark.parametrize("bf16", [False, True], ids=["x3", "bf16"])
@pytest.mark.parametrize("case", S2_CASES)
def test_conv2d_planes_stride2_forward(backend, case, bf16):
    """conv_planes_s2fwd_kernel (round 6): tf.nn.conv2d(strides 2, 'SAME') + bias + leaky (Nets/sharedLayers.py:54-66; DispNet conv2: Nets/DispNet.py:80-84) from
    pre-split planes, patch columns split by parity.  TF 'SAME' on even sizes pads (k - 2) // 2 in front (5x5 -> 1, 3x3 -> 0).  Split-bf16 against the fp64
    oracle at the 2^-16 level, plain bf16 against the oracle on bf16-rounded operands; ragged tiles (rows not a multiple of the tile, columns not a multiple of 32),
    output planes == the split of the fp32 result."""
    B, H, W, Ci, Co, k = case
    lib, dev = backend.lib, backend.device
    if bf16 and k != 5:
        pytest.skip("the one-plane form is instantiated for the 5x5 layers only")
    if not bf16 and Ci == 145:
        pytest.skip("conv3: one-plane instance only")
    x = _rand((B, H, W, Ci), 311, dev)
    w = _rand((k, k, Ci, Co), 312, dev, 0.1)
    b = _rand((Co,), 313, dev)
    keep = []
    assert ops.conv2d_planes_ok(lib, ops.view(x), w, 1, bf16=bf16, stride=2)
    assert not ops.conv2d_planes_ok(lib, ops.View(x, B, H - 1, W, Ci, Ci), w, 1, bf16=bf16, stride=2)          # odd sizes pad differently: not served
    xp = _planes_of(lib, x, dev, keep)
    planes = 1 if bf16 else 2
    bank = torch.full((ops.pack_bytes(w, planes, 2) // 4,), float("nan"), device=dev)
    assert ops.pack_bytes(w, 2, 2) == lib.pack32_bytes(k * k, Ci, Co)
    ops.pack_weights(lib, [(w, bank, planes, 2)], dev, keep)
    Ho, Wo = H // 2, W // 2
    y = torch.full((B, Ho, Wo, Co), float("nan"), device=dev)
    yp = ops.Planes(ops.Shadow(B, Ho, Wo, Co, dev), dev)
    ops.conv2d_planes(lib, xp, w, bank, b, out=ops.view(y), out_planes=yp, alpha=0.1, bf16=bf16, stride=2)
    assert "conv_planes_s2fwd_kernel<%dx%d" % (k, k) in lib.last_kernel().decode()
    backend.sync()
    if bf16:
        bf = lambda t: t.to(torch.bfloat16).float()
        ref = T.conv2d(bf(x.cpu()).double(), bf(w.cpu()).double(), b.cpu().double(), stride=2, alpha=0.1).float(); tol = 2e-5
    else:
        ref = T.conv2d(x.cpu().double(), w.cpu().double(), b.cpu().double(), stride=2, alpha=0.1).float(); tol = 4e-5
    assert ref.shape == y.shape
    err = (y.cpu() - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), err
    hi, lo = _split_ref(y.cpu())
    assert torch.equal(yp.hi.t.cpu()[..., :Co], hi) and torch.equal(yp.lo.t.cpu()[..., :Co], lo)
