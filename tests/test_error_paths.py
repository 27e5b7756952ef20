"""Error behaviour of the C-ABI (SURVEY 8(b) 'Errors'): int status codes (0 ok, negative = argument check failed and NOTHING
was launched or written, positive = hipError_t), message through mh_last_error(), no exception across the ABI; the Python
binding turns a non-zero status into MadnetHipError.  Runs on the CPU emulator build and, with -m gpu, on the MI355X."""
import ctypes as C

import pytest
import torch

from madnet_hip import _ffi, ops

ERR_ARG, ERR_ALIGN, ERR_UNSUPPORTED = -1, -2, -3


def _raw(backend, name):
    return getattr(backend.lib, "_raw_mh_" + name)


def _msg(backend):
    return backend.lib.last_error().decode()


def test_conv_argument_checks(backend):
    dev = backend.device
    x = torch.zeros(1, 8, 8, 8, device=dev); w = torch.zeros(3, 3, 8, 16, device=dev); b = torch.zeros(16, device=dev)
    y = torch.full((1, 8, 8, 16), float("nan"), device=dev)
    conv = _raw(backend, "conv2d")
    P = lambda t: C.c_void_p(t.data_ptr())
    good = ops.conv_desc(1, 8, 8, 8, 8, 8, 16, 3, 3, 1, 1, 1, 1, 0, 0, 8, 16)
    assert conv(C.byref(good), P(x), P(w), P(b), P(y), None, None) == 0
    backend.sync()
    y.fill_(float("nan")); backend.sync()
    # null input
    assert conv(C.byref(good), None, P(w), P(b), P(y), None, None) == ERR_ARG and "null" in _msg(backend)
    # ld smaller than the channel count
    bad = ops.conv_desc(1, 8, 8, 8, 8, 8, 16, 3, 3, 1, 1, 1, 1, 0, 0, 4, 16)
    assert conv(C.byref(bad), P(x), P(w), P(b), P(y), None, None) == ERR_ARG and "ld" in _msg(backend)
    # non-positive dimension / bad stride
    bad = ops.conv_desc(0, 8, 8, 8, 8, 8, 16, 3, 3, 1, 1, 1, 1, 0, 0, 8, 16)
    assert conv(C.byref(bad), P(x), P(w), P(b), P(y), None, None) == ERR_ARG
    bad = ops.conv_desc(1, 8, 8, 8, 8, 8, 16, 3, 3, 0, 1, 1, 1, 0, 0, 8, 16)
    assert conv(C.byref(bad), P(x), P(w), P(b), P(y), None, None) == ERR_ARG
    # unknown arithmetic mode / unsupported mode combination / non power-of-two stride for the gradient form
    bad = ops.conv_desc(1, 8, 8, 8, 8, 8, 16, 3, 3, 1, 1, 1, 1, 0, 0, 8, 16, precision=7)
    assert conv(C.byref(bad), P(x), P(w), P(b), P(y), None, None) == ERR_ARG and "precision" in _msg(backend)
    bad = ops.conv_desc(1, 8, 8, 8, 8, 8, 16, 3, 3, 1, 1, 1, 1, 1, 0, 8, 16)
    assert conv(C.byref(bad), P(x), P(w), P(b), P(y), None, None) == ERR_UNSUPPORTED
    bad = ops.conv_desc(1, 8, 8, 24, 24, 8, 16, 3, 3, 3, 1, 1, 1, 1, 1, 8, 16)
    assert conv(C.byref(bad), P(x), P(w), P(b), P(y), None, None) == ERR_UNSUPPORTED and "power-of-two" in _msg(backend)
    backend.sync()
    assert torch.isnan(y).all()                       # a failed check wrote nothing
    # the Python binding raises
    with pytest.raises(_ffi.MadnetHipError):
        backend.lib.conv2d(C.byref(good), None, P(w), P(b), P(y), None, None)


def test_corr_and_misc_argument_checks(backend):
    dev = backend.device
    P = lambda t: C.c_void_p(t.data_ptr())
    L = torch.zeros(1, 4, 16, 32, device=dev); R = torch.zeros_like(L)
    out = torch.full((1, 4, 16, 8), float("nan"), device=dev)
    corr = _raw(backend, "corr_fwd")
    # out_ld too small for D = 5 at channel offset 4
    assert corr(P(L), 32, P(R), 32, None, P(out), 8, 4, 1, 4, 16, 32, 2, 1, 0, 0, None) == ERR_ARG and "out_ld" in _msg(backend)
    # channel count not a multiple of 4 / misaligned pointer
    assert corr(P(L), 30, P(R), 30, None, P(out), 8, 0, 1, 4, 16, 30, 2, 1, 0, 0, None) == ERR_ALIGN
    assert corr(C.c_void_p(L.data_ptr() + 4), 32, P(R), 32, None, P(out), 8, 0, 1, 4, 16, 32, 2, 1, 0, 0, None) == ERR_ALIGN
    # copy_left needs coff >= C
    assert corr(P(L), 32, P(R), 32, None, P(out), 8, 0, 1, 4, 16, 32, 2, 1, 1, 0, None) == ERR_ALIGN
    backend.sync()
    assert torch.isnan(out).all()
    # filter-gradient workspace protocol: launch without a queried split count / with a wrong one
    x = torch.zeros(1, 8, 8, 8, device=dev); dz = torch.zeros(1, 8, 8, 16, device=dev); ws = torch.zeros(9 * 8 * 16 * 4, device=dev)
    d = ops.conv_desc(1, 8, 8, 8, 8, 8, 16, 3, 3, 1, 1, 1, 1, 0, 0, 8, 16)
    part = _raw(backend, "conv2d_wgrad_partial")
    n = C.c_int32(0)
    assert part(C.byref(d), P(x), P(dz), 16, P(ws), C.byref(n), None, None) == ERR_ARG and "query" in _msg(backend)
    assert part(C.byref(d), P(x), P(dz), 16, None, C.byref(n), None, None) == 0 and n.value >= 1        # the query itself
    wrong = C.c_int32(n.value + 3)
    assert part(C.byref(d), P(x), P(dz), 16, P(ws), C.byref(wrong), None, None) == ERR_ARG and "split count" in _msg(backend)
    assert _raw(backend, "wgrad_reduce")(None, 0, 0, None) == ERR_ARG
    # reflect pad larger than the image, loss on a 2x2 image, momentum on nothing
    img = torch.zeros(1, 4, 4, 3, device=dev); o = torch.zeros(1, 64, 64, 4, device=dev)
    assert _raw(backend, "pad_reflect")(P(img), P(o), 1, 4, 4, 3, 64, 64, 30, 30, 4, 1.0, 0.0, None) == ERR_ARG and "REFLECT" in _msg(backend)
    z = torch.zeros(64, device=dev)
    assert _raw(backend, "reprojection_loss")(P(z), P(z), P(z), P(z), P(z), None, 1.0, 1, 2, 2, None) == ERR_ARG
    assert _raw(backend, "momentum")(P(z), P(z), P(z), 0, 1e-4, 0.9, 1.0, None) == ERR_ARG
    # offline-training entry points: Adam without its beta-power state / with beta = 1, supervised loss with max_disp <= 0
    assert _raw(backend, "adam")(P(z), P(z), P(z), P(z), 64, None, 1e-4, 0.9, 0.999, 1e-8, 1.0, None) == ERR_ARG
    assert _raw(backend, "adam")(P(z), P(z), P(z), P(z), 64, P(z), 1e-4, 1.0, 0.999, 1e-8, 1.0, None) == ERR_ARG and "beta" in _msg(backend)
    assert _raw(backend, "adam_advance")(None, 0.9, 0.999, None) == ERR_ARG
    assert _raw(backend, "supervised_loss")(P(z), P(z), P(z), P(z), None, 1.0, 1.0, 0.0, 1, 2, 2, None) == ERR_ARG and "max_disp" in _msg(backend)
    # plan executor: unknown op kind / lane out of range, reported with the op index
    op = (_ffi.Op * 1)(); op[0].kind = 99
    assert _raw(backend, "plan_run")(op, 1, None) == ERR_ARG and "plan op 0" in _msg(backend)
    op[0].kind = _ffi.OP_FILL; op[0].i[26] = 7
    assert _raw(backend, "plan_run")(op, 1, None) == ERR_ARG and "lane" in _msg(backend)


def test_round3_entry_points_argument_checks(backend):
    """mh_head_bwd, mh_conv2d_head, mh_conv2d_sh3, mh_conv2d_takes_shadows, mh_plans_run: argument errors return a negative code with a message
    and launch nothing; the dispatch query answers 1 only for a launch that would stage its input's shadow."""
    dev = backend.device
    P = lambda t: C.c_void_p(t.data_ptr())
    B, H, W, N = 1, 12, 20, 32
    w = torch.zeros(3, 3, N, 1, device=dev)
    dV = torch.full((B, H, W), float("nan"), device=dev)
    dx = torch.full((B, H, W, N), float("nan"), device=dev)
    du = torch.zeros(B, 2 * H, 2 * W, device=dev)
    hb = _raw(backend, "head_bwd")
    d = _ffi.HeadBwdDesc()
    d.kind, d.B, d.H, d.W, d.N, d.Hr, d.Wr, d.Ho, d.Wo, d.mul, d.dx_ld = 0, B, H, W, N, 2 * H, 2 * W, 2 * H, 2 * W, 1.0, N
    assert hb(C.byref(d), None, None, P(dV), None, P(w), P(dx), None, None, None) == ERR_ARG            # no source at all
    d.kind = 2
    assert hb(C.byref(d), P(du), None, P(dV), None, P(w), P(dx), None, None, None) == ERR_ARG and "kind" in _msg(backend)
    d.kind, d.N = 0, 30
    assert hb(C.byref(d), P(du), None, P(dV), None, P(w), P(dx), None, None, None) == ERR_ARG             # N must be a multiple of 4
    d.N, d.dx_ld = N, N + 2
    assert hb(C.byref(d), P(du), None, P(dV), None, P(w), P(dx), None, None, None) == ERR_ALIGN
    backend.sync()
    assert torch.isnan(dV).all() and torch.isnan(dx).all()
    # mh_conv2d_head: only a forward conv with ONE output channel
    x = torch.zeros(B, H, W, N, device=dev); o = torch.zeros(B, H, W, device=dev); o2 = torch.zeros(B, H, W, device=dev)
    dc = ops.conv_desc(B, H, W, H, W, N, 8, 3, 3, 1, 1, 1, 1, 0, 0, N, 8)
    head = _raw(backend, "conv2d_head")
    assert head(C.byref(dc), P(x), P(w), None, P(o), P(o2), 1, None, 0, None) == ERR_ARG and "ONE output channel" in _msg(backend)
    dc1 = ops.conv_desc(B, H, W, H, W, N, 1, 3, 3, 1, 1, 1, 1, 0, 0, N, 1)
    assert head(C.byref(dc1), P(x), P(w), None, P(o), P(o2), 0, None, 0, None) == ERR_ARG                 # pixel stride of an extra output
    # mh_conv2d_sh3: shadow-only needs the shadow and no accumulation
    wd = torch.zeros(3, 3, 64, 64, device=dev); gz = torch.zeros(B, H, W, 64, device=dev); gx = torch.zeros(B, H, W, 64, device=dev)
    dd = ops.conv_desc(B, H, W, H, W, 64, 64, 3, 3, 1, 1, 1, 1, 1, 1, 64, 64, precision=1)
    sh3 = _raw(backend, "conv2d_sh3")
    assert sh3(C.byref(dd), P(gz), None, P(wd), None, None, P(gx), None, None, None, 1, None) == ERR_ARG and "SHADOW_ONLY" in _msg(backend)
    # the dispatch query: the patch-staged input gradient (forced on at this size) stages shadows, the same layer in fp32 does not
    q = _raw(backend, "conv2d_takes_shadows")
    backend.lib.tune_conv_patch(128)
    try:
        assert q(C.byref(dd), P(gz), P(wd), None, P(gx), None) == 1
        d0 = ops.conv_desc(B, H, W, H, W, 64, 64, 3, 3, 1, 1, 1, 1, 1, 1, 64, 64, precision=0)
        assert q(C.byref(d0), P(gz), P(wd), None, P(gx), None) == 0
        df = ops.conv_desc(B, H, W, H, W, 64, 64, 3, 3, 1, 1, 1, 1, 0, 0, 64, 64, precision=1)
        assert q(C.byref(df), P(gz), P(wd), None, P(gx), None) == 0                                       # a forward conv
    finally:
        backend.lib.tune_conv_patch(-1)
    # mh_plans_run: a plan that uses the last lane (reserved for the branch streams) is refused
    op = (_ffi.Op * 1)(); op[0].kind = _ffi.OP_FILL; op[0].i[26] = 4
    refs = (_ffi.PlanRef * 1)(); refs[0].ops, refs[0].nops = C.addressof(op), 1
    assert _raw(backend, "plans_run")(refs, 1, None) < 0 and "lane" in _msg(backend)
    assert _raw(backend, "plans_run")(refs, 0, None) == ERR_ARG
