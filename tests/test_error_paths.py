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
