"""Streaming filter-gradient kernel (csrc/wgrad_stream.hip: mh_shadow_cast + mh_wgrad_stream_plan + mh_wgrad_stream) against the fp64 oracle
on bf16-rounded operands -- the Conv2DBackpropFilter / BiasAddGrad nodes of Stereo_Online_Adaptation.py:126-128 for the stride-1 3x3 layers
(Nets/sharedLayers.py:54-77)."""
import ctypes as C

import pytest
import torch

from madnet_hip import _ffi, ops
from oracle import tf_ops as T


def _rand(shape, seed, dev, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _padded(x, ld):
    B, H, W, Cc = x.shape
    buf = torch.zeros(B, H, W, ld, device=x.device)
    buf[..., :Cc] = x
    return buf, ops.View(buf, B, H, W, Cc, ld)


def _oracle(x, gz, dil):
    Ci, Co = x.shape[-1], gz.shape[-1]
    w0 = torch.zeros(3, 3, Ci, Co, dtype=torch.float64, requires_grad=True)
    y = T.conv2d(_bf(x.cpu()).double(), w0, None, dilation=dil, alpha=1.0)
    (gw,) = torch.autograd.grad(y, [w0], _bf(gz.cpu()).double())
    return gw, _bf(gz.cpu()).double().sum((0, 1, 2))


def test_shadow_cast(backend):
    """fp32 NHWC (vector and ragged channel counts, padded source rows) -> bf16 NHWC with the channel stride rounded up to 32 and a ZERO pad."""
    dev = backend.device
    pairs, srcs = [], []
    for k, (B, H, W, Cc, ld) in enumerate([(1, 5, 7, 32, 32), (2, 3, 9, 38, 40), (1, 4, 6, 1, 1), (1, 2, 3, 70, 72), (1, 3, 5, 96, 96)]):
        x = _rand((B, H, W, Cc), 40 + k, dev)
        buf, v = _padded(x, ld)
        if ld != Cc:
            buf[..., Cc:] = 9.0                   # a concat neighbour: must not leak into the shadow's padding
        sh = ops.Shadow(B, H, W, Cc, dev)
        sh.t.fill_(3.0)                            # stale contents
        pairs.append((v, sh)); srcs.append(x)
    keep = []
    ops.shadow_cast(backend.lib, pairs, dev, keep)
    backend.sync()
    for (v, sh), x in zip(pairs, srcs):
        got = sh.t.float().cpu()
        assert sh.ld % 32 == 0 and got.shape[-1] == sh.ld
        assert torch.equal(got[..., :sh.C], _bf(x.cpu())), "round-to-nearest-even bf16 of the source"
        assert (got[..., sh.C:] == 0).all(), "padding channels must be zero"


def _oracle_s2(x, gz):
    Ci, Co = x.shape[-1], gz.shape[-1]
    w0 = torch.zeros(3, 3, Ci, Co, dtype=torch.float64, requires_grad=True)
    y = T.conv2d(_bf(x.cpu()).double(), w0, None, stride=2, alpha=1.0)
    (gw,) = torch.autograd.grad(y, [w0], _bf(gz.cpu()).double())
    return gw, _bf(gz.cpu()).double().sum((0, 1, 2))


def _run_batch(backend, layers, target_wgs, nwaves, stride=1):
    """layers: [(B, H, W, Cin, Cout, dil, in_ld)] (H, W = the INPUT size) -> [(dw, db)] through shadow_cast + wgrad_stream + wgrad_reduce."""
    dev = backend.device
    items, pairs, outs, refs = [], [], [], []
    for k, (B, H, W, Ci, Co, dil, ild) in enumerate(layers):
        x = _rand((B, H, W, Ci), 700 + 2 * k, dev)
        gz = _rand((B, H // stride, W // stride, Co), 701 + 2 * k, dev)
        xb, xv = _padded(x, ild)
        if ild != Ci:
            xb[..., Ci:] = 7.5
        xs, zs = ops.Shadow(B, H, W, Ci, dev), ops.Shadow(B, H // stride, W // stride, Co, dev)
        pairs += [(xv, xs), (ops.view(gz) if Co > 1 else ops.view(gz[..., 0].contiguous()), zs)]
        dw = torch.full((3, 3, Ci, Co), float("nan"), device=dev)
        db = torch.zeros(Co, device=dev)
        items.append((xs, zs, dw, db, dil))
        outs.append((dw, db)); refs.append(_oracle(x, gz, dil) if stride == 1 else _oracle_s2(x, gz))
    wsa = ops.WgradWorkspace(dev); segs, keep = [], []
    ops.shadow_cast(backend.lib, pairs, dev, keep)
    ops.wgrad_stream(backend.lib, backend.lib, wsa, segs, items, dev, keep, target_wgs=target_wgs, nwaves=nwaves)
    name = backend.lib.last_kernel().decode()
    ops.wgrad_reduce(backend.lib, segs, dev, keep)
    backend.sync()
    return outs, refs, name, segs


STREAM_BATCHES = [
    # (layers, target workgroups, waves per workgroup)
    ([(1, 9, 40, 32, 32, 1, 32)], 4, 4),                                    # one tile, 2 strips (32 + 8 columns), several runs per wave
    ([(1, 12, 70, 38, 64, 1, 40), (1, 12, 70, 64, 1, 1, 64)], 12, 4),       # ragged K (38 in a 40-wide row), N = 1 head, ragged strip (70 = 2 x 32 + 6)
    ([(2, 7, 33, 32, 40, 1, 32)], 6, 8),                                    # batch 2, N = 40 (two column tiles, the second 8 wide), 8 waves
    ([(1, 16, 36, 32, 32, 2, 32), (1, 16, 36, 40, 32, 4, 40)], 8, 4),       # dilation 2 and 4 in one launch
    ([(1, 21, 34, 32, 32, 8, 32)], 3, 4),                                   # dilation 8, H % d != 0 (rows past the image are zeros)
    ([(1, 34, 40, 32, 32, 16, 32)], 4, 4),                                  # dilation 16: 64-pixel row slots (NXG = 4)
    ([(1, 6, 20, 64, 64, 1, 64)], 256, 8),                                  # fewer rows than waves: idle waves, >= 2 rows per wave floor
]


@pytest.mark.parametrize("batch", STREAM_BATCHES)
def test_wgrad_stream(backend, batch):
    layers, wgs, nw = batch
    outs, refs, name, segs = _run_batch(backend, layers, wgs, nw)
    assert "wgrad_stream_kernel" in name, name
    for (dw, db), (gw, gb), lay in zip(outs, refs, layers):
        sc = max(1.0, gw.abs().max().item())
        err = (dw.cpu().double() - gw).abs().max().item()
        assert err <= 2e-5 * sc, (lay, err, sc)
        assert (db.cpu().double() - gb).abs().max().item() <= 1e-4 * max(1.0, gb.abs().max().item()), lay


STREAM_S2_BATCHES = [
    ([(1, 12, 64, 16, 32, 1, 16)], 4, 4),                                   # the pyramid's 16 -> 32 down-sampling layer: one 32-column strip of the output
    ([(2, 16, 80, 32, 64, 1, 32), (2, 8, 40, 64, 96, 1, 64)], 9, 4),         # two towers, ragged strips (40 = 32 + 8; 20 columns), two layers in one launch
    ([(1, 20, 72, 40, 33, 1, 40)], 6, 5),                                   # ragged channel counts on both sides, 5 waves
]


@pytest.mark.parametrize("batch", STREAM_S2_BATCHES)
def test_wgrad_stream_stride2(backend, batch):
    """the stride-2 instance (the pyramid's down-sampling layers): output pixel (y, x) reads input rows 2y .. 2y + 2, columns 2x .. 2x + 2 ('SAME' on
    even sizes pads behind only) -- two new 65-pixel input rows per step, pixel stride 2 in the transposing reads"""
    layers, wgs, nw = batch
    outs, refs, name, segs = _run_batch(backend, layers, wgs, nw, stride=2)
    assert "wgrad_stream_kernel<5,1,s2>" in name, name
    for (dw, db), (gw, gb), lay in zip(outs, refs, layers):
        sc = max(1.0, gw.abs().max().item())
        err = (dw.cpu().double() - gw).abs().max().item()
        assert err <= 2e-5 * sc, (lay, err, sc)
        assert (db.cpu().double() - gb).abs().max().item() <= 1e-4 * max(1.0, gb.abs().max().item()), lay


def test_wgrad_stream_mixed_strides_one_launch(backend):
    """a pyramid batch -- two down-sampling layers and the stride-1 layers behind them -- in ONE launch (wgrad_stream_mixed_kernel: the workgroup takes
    the instance of its layer's stride)"""
    dev = backend.device
    specs = [(2, 16, 64, 16, 32, 2), (2, 8, 32, 32, 32, 1), (2, 8, 32, 32, 64, 2), (2, 4, 16, 64, 64, 1)]      # (B, H, W, Cin, Cout, stride), H x W = input size
    items, pairs, outs, refs = [], [], [], []
    for k, (B, H, W, Ci, Co, st) in enumerate(specs):
        x = _rand((B, H, W, Ci), 900 + 2 * k, dev); gz = _rand((B, H // st, W // st, Co), 901 + 2 * k, dev)
        xs, zs = ops.Shadow(B, H, W, Ci, dev), ops.Shadow(B, H // st, W // st, Co, dev)
        pairs += [(ops.view(x), xs), (ops.view(gz), zs)]
        dw = torch.full((3, 3, Ci, Co), float("nan"), device=dev); db = torch.zeros(Co, device=dev)
        items.append((xs, zs, dw, db, 1)); outs.append((dw, db))
        refs.append(_oracle(x, gz, 1) if st == 1 else _oracle_s2(x, gz))
    wsa = ops.WgradWorkspace(dev); segs, keep = [], []
    ops.shadow_cast(backend.lib, pairs, dev, keep)
    ops.wgrad_stream(backend.lib, backend.lib, wsa, segs, items, dev, keep, target_wgs=16, nwaves=4)
    assert "wgrad_stream_mixed_kernel layers 4" in backend.lib.last_kernel().decode()
    ops.wgrad_reduce(backend.lib, segs, dev, keep)
    backend.sync()
    for (dw, db), (gw, gb), sp in zip(outs, refs, specs):
        sc = max(1.0, gw.abs().max().item())
        assert (dw.cpu().double() - gw).abs().max().item() <= 2e-5 * sc, sp
        assert (db.cpu().double() - gb).abs().max().item() <= 1e-4 * max(1.0, gb.abs().max().item()), sp


@pytest.mark.gpu
def test_wgrad_stream_stride2_pyramid_sizes(hip):
    """conv3 / conv5 / conv7 of the pyramid at the headline size, both towers, one launch"""
    layers = [(2, 192, 640, 16, 32, 1, 16), (2, 96, 320, 32, 64, 1, 32), (2, 48, 160, 64, 96, 1, 64)]
    for rep in range(2):
        outs, refs, name, segs = _run_batch(hip, layers, 256, 4, stride=2)
        for (dw, db), (gw, gb), lay in zip(outs, refs, layers):
            sc = max(1.0, gw.abs().max().item())
            assert (dw.cpu().double() - gw).abs().max().item() <= 2e-5 * sc, (lay, name)
            assert (db.cpu().double() - gb).abs().max().item() <= 1e-4 * max(1.0, gb.abs().max().item()), lay


@pytest.mark.gpu
@pytest.mark.parametrize("dist", [1, 2])
def test_wgrad_stream_full_size_context_batch(hip, dist):
    """The context network's six streamed layers at 96 x 320 in one launch (dilations 1 .. 16 -> 64-pixel slots), 256 workgroups; both prefetch
    distances of the d <= 8 instance on the estimator batch (counted vmcnt waits are only exercised on the hardware: the emulator's DMA is synchronous)."""
    H, W = 96, 320
    est = [(1, H, W, 38, 128, 1, 40), (1, H, W, 128, 128, 1, 128), (1, H, W, 128, 96, 1, 128), (1, H, W, 96, 64, 1, 96), (1, H, W, 64, 32, 1, 64),
           (1, H, W, 32, 1, 1, 32)]
    ctx = [(1, H, W, 33, 128, 1, 36), (1, H, W, 128, 128, 2, 128), (1, H, W, 128, 128, 4, 128), (1, H, W, 128, 96, 8, 128), (1, H, W, 96, 64, 16, 96),
           (1, H, W, 64, 32, 1, 64), (1, H, W, 32, 1, 1, 32)]
    hip.lib.tune_wgrad_stream(dist)
    try:
        for layers, nw in ((est, 8 if dist == 1 else 6), (ctx, 7)):
            for rep in range(2):                   # twice: a wrong vmcnt count shows as run-to-run differences
                outs, refs, name, segs = _run_batch(hip, layers, 256, nw)
                for (dw, db), (gw, gb), lay in zip(outs, refs, layers):
                    sc = max(1.0, gw.abs().max().item())
                    assert (dw.cpu().double() - gw).abs().max().item() <= 2e-5 * sc, (lay, name)
                    assert (db.cpu().double() - gb).abs().max().item() <= 1e-4 * max(1.0, gb.abs().max().item()), lay
    finally:
        hip.lib.tune_wgrad_stream(0)


def test_wgrad_stream_plan_divides_the_workgroups(backend):
    """The host-side planner: split counts proportional to the rows, grid <= target, >= 2 rows per wave, blk0 = exclusive prefix."""
    arr = (_ffi.WgsLayer * 3)()
    for L, (K, N, d) in zip(arr, [(128, 128, 1), (64, 32, 1), (96, 64, 16)]):
        L.B, L.H, L.W, L.K, L.N, L.dil, L.x_ld, L.dz_ld = 1, 96, 320, K, N, d, ops.shadow_ld(K), ops.shadow_ld(N)
    nb = C.c_int32(0)
    backend.lib.wgrad_stream_plan(arr, 3, 256, 8, C.byref(nb))
    tiles = [L.ktiles * L.ntiles for L in arr]
    assert tiles == [16, 2, 6]
    assert nb.value == sum(t * L.splits for t, L in zip(tiles, arr)) and nb.value <= 256
    assert arr[0].blk0 == 0 and arr[1].blk0 == tiles[0] * arr[0].splits and arr[2].blk0 == arr[1].blk0 + tiles[1] * arr[1].splits
    assert arr[0].splits == arr[1].splits == arr[2].splits >= 8          # same rows (960) everywhere: same split count
    nb2 = C.c_int32(0)
    backend.lib.wgrad_stream_plan(arr, 3, 100000, 8, C.byref(nb2))
    assert all(L.splits * 16 <= 960 + 15 for L in arr), "at least two rows per wave"
    with pytest.raises(_ffi.MadnetHipError):
        arr[0].x_ld = 96
        backend.lib.wgrad_stream_plan(arr, 3, 256, 8, C.byref(nb))
