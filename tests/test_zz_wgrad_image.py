"""The image layer's filter gradient on its own kernel (wgrad_image_kernel, csrc/wgrad.hip; mh_tune_wgrad_image: the default since round 4, 256 workgroups) and the
momentum update per filter-gradient batch on the batch's lane (Schedule.EARLY_UPDATE: measured in round 5, left OFF for MADNet -- profiles/r05_experiments.txt #9 --,
ON for DispNet).  Both were written after round 3's GPU budget was spent; this file (the last one pytest collects) keeps their emulator and MI355X parity tests."""
import pytest
import torch

from madnet_hip import ops, engine as E, synthetic as S
from oracle import madnet as OM
from test_conv_parity import _rand, _padded, _oracle_grads, _bf
from test_engine_parity import _backend, _ffi_mod


# (B, H, W, Cin, in_ld, stride): the image layer's filter gradient (3x3, Cout = 16) on its own kernel (mh_tune_wgrad_image, default off)
IMAGE_WGRAD_CASES = [(2, 96, 192, 3, 3, 2), (1, 96, 96, 3, 4, 1), (2, 131, 143, 1, 1, 2), (1, 181, 207, 2, 4, 2)]       # >= 8192 output pixels, output rows of 8 k pixels


@pytest.mark.parametrize("how", ["single", "plan"])
@pytest.mark.parametrize("case", IMAGE_WGRAD_CASES)
def test_wgrad_image_layer_kernel(backend, case, how):
    """wgrad_image_kernel (one partial gradient per workgroup, exact-fp32 bias gradient) against the oracle on bf16-rounded operands and against the tiled
    kernel it replaces; also as an op of a recorded plan (mh_plan_run batches partial filter gradients: this kernel must fall out of the grouped grid)."""
    from madnet_hip import plan as PL
    B, H, W, Ci, ld, s = case
    dev = backend.device
    x = _rand((B, H, W, Ci), 301, dev) * 50.0
    Ho, Wo, _, _ = ops.conv_geometry(H, W, 3, 3, s, 1)
    gz = _rand((B, Ho, Wo, 16), 302, dev)
    xb, xv = _padded(x, ld)
    if ld > Ci:
        xb[..., Ci:] = float("inf")            # row padding is never an operand
    w0 = torch.zeros(3, 3, Ci, 16); b0 = torch.zeros(16)
    _, _, gw_ref, _ = _oracle_grads(_bf(x.cpu()), w0, b0, s, 1, 1.0, _bf(gz.cpu()))
    _, _, _, gb_ref = _oracle_grads(x.cpu(), w0, b0, s, 1, 1.0, gz.cpu())
    res = {}
    for on in (1, 0):
        prev = backend.lib.tune_wgrad_image(on)
        ops.PRECISION_BWD = 1
        ops.PRECISION = 1
        try:
            wsa = ops.WgradWorkspace(dev); wsa.CHUNK = 1 << 20
            rec = PL.Recorder() if how == "plan" else None
            tgt = rec if rec is not None else backend.lib
            segs, keep = [], []
            dw = torch.full((3, 3, Ci, 16), float("nan"), device=dev); db = torch.zeros(16, device=dev)
            ops.conv2d_wgrad_partial(tgt, backend.lib, wsa, segs, xv, ops.view(gz), dw, db, stride=s)
            # a second, ordinary layer in the same batch (grouped launch of the plan executor)
            x2 = _rand((1, 12, 20, 32), 303, dev); gz2 = _rand((1, 12, 20, 32), 304, dev)
            dw2 = torch.full((3, 3, 32, 32), float("nan"), device=dev); db2 = torch.zeros(32, device=dev)
            ops.conv2d_wgrad_partial(tgt, backend.lib, wsa, segs, ops.view(x2), ops.view(gz2), dw2, db2)
            assert Wo % 8 == 0 and segs[0][3] == (min(64, (B * Ho * Wo + 511) // 512) if on else segs[0][3])
            ops.wgrad_reduce(tgt, segs, dev, keep)
            if rec is not None:
                rec.compile().run(backend.lib, None)
            backend.sync()
        finally:
            ops.PRECISION = 0
            ops.PRECISION_BWD = None
            backend.lib.tune_wgrad_image(prev)
        res[on] = (dw.cpu().clone(), db.cpu().clone(), dw2.cpu().clone())
    _, _, gw32, _ = _oracle_grads(x.cpu(), w0, b0, s, 1, 1.0, gz.cpu())
    tol = 1e-4 * max(1.0, gw_ref.abs().max().item())
    assert (res[1][0] - gw_ref).abs().max().item() <= tol
    # the tiled path it replaces: bf16 operands on 16-byte rows, exact fp32 (scalar loader) on 3- / 1-float rows
    assert min((res[0][0] - gw_ref).abs().max().item(), (res[0][0] - gw32).abs().max().item()) <= tol
    for on in (1, 0):
        assert (res[on][1] - gb_ref).abs().max().item() <= 1e-4 * max(1.0, gb_ref.abs().max().item()), on
    assert torch.equal(res[1][2], res[0][2])                      # the neighbour layer is untouched by the switch


def test_wgrad_image_layer_kernel_declines_ragged_rows(backend):
    """output rows that are not a multiple of 8 pixels (a lane's 8 pixels would straddle rows) and small layers stay on the tiled kernel"""
    import ctypes as C
    dev = backend.device
    prev = backend.lib.tune_wgrad_image(1)
    ops.PRECISION_BWD = 1
    ops.PRECISION = 1
    try:
        for (H, W, s) in ((96, 180, 2), (40, 64, 1)):
            x = _rand((2, H, W, 3), 311, dev)
            Ho, Wo, _, _ = ops.conv_geometry(H, W, 3, 3, s, 1)
            gz = _rand((2, Ho, Wo, 16), 312, dev)
            xb, xv = _padded(x, 4)
            wsa = ops.WgradWorkspace(dev); wsa.CHUNK = 1 << 20
            segs = []
            dw = torch.zeros(3, 3, 3, 16, device=dev); db = torch.zeros(16, device=dev)
            ops.conv2d_wgrad_partial(backend.lib, backend.lib, wsa, segs, xv, ops.view(gz), dw, db, stride=s)
            assert "wgrad_image" not in backend.lib.last_kernel().decode()
    finally:
        ops.PRECISION = 0
        ops.PRECISION_BWD = None
        backend.lib.tune_wgrad_image(prev)


def _image_layer_kernel_ab(backend, H, W):
    """FULL 'mixed' step with conv1's filter gradient on wgrad_image_kernel (mh_tune_wgrad_image(1) while the plan is recorded AND run) against the
    default plan: conv1's gradient within the bf16 summation-order noise, every other gradient unchanged."""
    F = _ffi_mod()
    wn = S.calibrated_weights(OM.variable_shapes(), 1)
    l, r, gt = S.make_pair(H, W)
    out = {}
    for on in (1, 0):
        prev = backend.lib.tune_wgrad_image(on)
        try:
            eng = E.MadNetEngine(backend.lib, H, W, B=1, device=backend.device, weights=wn, precision="mixed")
            eng.set_inputs(l, r, gt[..., 0])
            plan = eng.build_plan("FULL", lr=1e-4, update=False)
            splits = [o.i[23] for o in plan.arr if o.kind == F.OP_WGRAD_PARTIAL and o.i[5] == 3]
            plan.run(backend.lib, 0)
            backend.sync()
        finally:
            backend.lib.tune_wgrad_image(prev)
        out[on] = (eng.params.g.clone().cpu(), eng.params.tensor("model/gc-read-pyramid/conv1/weights", "g").clone().cpu(), splits)
    M = 2 * (eng.Hp // 2) * (eng.Wp // 2)
    assert out[1][2] == [min(64, (M + 511) // 512)] and out[0][2] != out[1][2], (out[1][2], out[0][2], M)
    d1 = (out[1][1] - out[0][1]).abs().max().item()
    assert d1 <= 2e-4 * out[0][1].abs().max().item(), d1
    assert ((out[1][0] - out[0][0]).norm() / out[0][0].norm()).item() <= 1e-4


def test_step_with_image_layer_filter_gradient_kernel_emulated():
    from conftest import _emul_backend
    _image_layer_kernel_ab(_emul_backend(), 120, 180)          # pads to 128 x 192: conv1 writes 2 x 64 x 96 = 12288 pixels


@pytest.mark.gpu
def test_step_with_image_layer_filter_gradient_kernel_gpu():
    _image_layer_kernel_ab(_backend("hip"), 375, 1242)




def _early_update_ab(backend, H, W, precision):
    """FULL momentum step with the per-batch updates (Schedule.EARLY_UPDATE) against the single update behind the join: the same elementwise update of the
    same gradients; every parameter updated exactly once; only the last batch's layers + whatever has no filter gradient are left for the final launch."""
    F = _ffi_mod()
    wn = S.calibrated_weights(OM.variable_shapes(), 1)
    l, r, gt = S.make_pair(H, W)
    res = {}
    try:
        for early in (True, False):
            eng = E.MadNetEngine(backend.lib, H, W, B=1, device=backend.device, weights=wn, precision=precision, schedule=E.Schedule(EARLY_UPDATE=early))
            eng.set_inputs(l, r, gt[..., 0])
            plan = eng.build_plan("FULL", lr=1e-3)
            P = eng.params
            mom = [o for o in plan.arr if o.kind == F.OP_MOMENTUM]
            spans = sorted(((o.p[0] - P.w.data_ptr()) // 4, (o.p[0] - P.w.data_ptr()) // 4 + o.n) for o in mom)
            assert spans[0][0] == 0 and spans[-1][1] == P.total and all(a[1] == b[0] for a, b in zip(spans, spans[1:])), spans       # a partition
            on_side = sum(1 for o in mom if (o.i[26] & 0xff) > 0)
            assert (len(mom) >= 6 and on_side >= len(mom) - 1) if early else (len(mom) == 1 and on_side == 0), (len(mom), on_side)
            snaps = []
            for _ in range(2):
                plan.run(backend.lib, 0)
                backend.sync()
                snaps.append((P.w.clone().cpu(), P.m.clone().cpu(), eng.pred.clone().cpu()))
            res[early] = snaps
    finally:
        pass
    # two runs of the SAME plan differ by ~1e-8 in a few gradient elements (fp32 atomics), amplified by the bf16 roundings of the second step
    for step, tols in ((0, (1e-9, 1e-7, 1e-5)), (1, (1e-6, 4e-4, 1e-3))):
        for a, b, tol in zip(res[True][step], res[False][step], tols):
            assert (a - b).abs().max().item() <= tol, (step, tol, (a - b).abs().max().item())


def test_early_update_equals_late_update_emulated():
    from conftest import _emul_backend
    _early_update_ab(_emul_backend(), 60, 100, "mixed")


@pytest.mark.gpu
def test_early_update_equals_late_update_gpu():
    _early_update_ab(_backend("hip"), 375, 1242, "mixed")
