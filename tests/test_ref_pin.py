"""Pins the correlation path to the REFERENCE'S OWN kernel.

tests/golden/shift_corr_ref.npz holds what CorrelateData + ShiftCorrKernelLauncher
(/root/reference/Nets/Native/shift_corr.cu.cc:17-70,193-233) produce on seeded inputs (the source compiled where it lies by
oracle/Makefile; generator: tests/golden/make_shift_corr_golden.py).  Against those values:
  * not gpu : every oracle restatement of the correlation (torch oracle/tf_ops.py, loop-level oracle/loops.py, plain-C
              oracle/c/oracle_ops.c), and -- when oracle/_ref is built -- the reference kernel itself re-run live;
  * emul/hip: the product's mh_corr_fwd (NHWC) and the literal-signature drop-in mh_shift_corr (W-padded NHWC in, NCHW out);
  * gpu     : the reference kernel built for gfx950 (oracle/_ref/libshift_corr_ref.so) on the MI355X at the FULL MADNet /
              DispNet cost-volume shapes, three-way against mh_corr_fwd, mh_shift_corr and the oracle.
Forward only: the reference's backward kernels are defective (CorrelateDataBackward0/1 are fed in0 for
both operands by shift_corr.cc:76-77 and write NCHW offsets into NHWC-shaped outputs), so the gradient is pinned to the TF
formulation (tests/test_ops_parity.py, tests/test_oracle_crosscheck.py) and mh_shift_corr_grad to autograd of this forward.
"""
import ctypes as C
import importlib.util
import os
import subprocess
import zlib

import numpy as np
import pytest
import torch

from madnet_hip import ops
from oracle import loops as Lp
from oracle import tf_ops as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("make_shift_corr_golden", os.path.join(ROOT, "tests", "golden", "make_shift_corr_golden.py"))
G = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(G)
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "shift_corr_ref.npz"))
CASES = [pytest.param(c, id=c[0]) for c in G.CASES]
# fp32 sums of C products in a different order than the reference's 32 strided partial sums (shift_corr.cu.cc:47-63)
ATOL = 2e-6


def _inputs(case):
    name, B, H, W, Cc, md = case
    L, R = G.make_inputs(*case)
    crc = GOLD[name + "/crc"]
    assert zlib.crc32(L.tobytes()) == crc[0] and zlib.crc32(R.tobytes()) == crc[1], "input generator drifted from the fixture"
    return L, R, GOLD[name + "/out"].transpose(0, 2, 3, 1)          # golden as NHWC [B,H,W,D]


@pytest.mark.parametrize("case", CASES)
def test_oracles_match_reference_kernel(case):
    name, B, H, W, Cc, md = case
    L, R, gold = _inputs(case)
    o = T.correlation(torch.from_numpy(L), torch.from_numpy(R), md, 1).numpy()
    assert np.abs(o - gold).max() <= ATOL
    if H * W * Cc * (2 * md + 1) <= 2_000_000:                       # the python-loop restatement is slow
        assert np.abs(Lp.correlation(L, R, md, 1) - gold).max() <= ATOL
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "c/liboracle_ops.so"], check=True, capture_output=True)
    clib = C.CDLL(os.path.join(ROOT, "oracle", "c", "liboracle_ops.so"))
    out = np.zeros((B, H, W, 2 * md + 1), np.float32)
    fp = lambda a: a.ctypes.data_as(C.c_void_p)
    clib.oc_corr_fwd(fp(L), fp(R), fp(out), B, H, W, Cc, md, 1)
    assert np.abs(out - gold).max() <= ATOL


def test_reference_kernel_reproduces_fixture_live():
    """Where oracle/_ref exists (this container: built from /root/reference by oracle/Makefile; GPU box: shipped prebuilt),
    the reference's kernel -- CPU-emulated build -- is re-run and must reproduce the committed fixture bit for bit."""
    so = os.path.join(ROOT, "oracle", "_ref", "libshift_corr_ref_cpu.so")
    if os.path.exists("/root/reference/Nets/Native/shift_corr.cu.cc"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    lib = C.CDLL(so)
    for case in G.CASES:
        L, R, gold = _inputs(case)
        assert np.array_equal(G.run_reference(lib, L, R, case[5]).transpose(0, 2, 3, 1), gold), case[0]


def _padded(x, md, dev):
    return torch.from_numpy(np.ascontiguousarray(np.pad(x, ((0, 0), (0, 0), (md, md), (0, 0))))).to(dev)


@pytest.mark.parametrize("case", CASES)
def test_hip_corr_matches_reference_kernel(backend, case):
    name, B, H, W, Cc, md = case
    if backend.name == "emul" and md == 40:
        pytest.skip("81 shifts x 128 channels through the emulator: covered on the GPU")
    L, R, gold = _inputs(case)
    dev = backend.device
    Lt, Rt = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
    D = 2 * md + 1
    if Cc % 4 == 0:               # mh_corr_fwd: 16-byte rows only (every channel count of the two networks is a multiple of 4)
        out = torch.full((B, H, W, D), float("nan"), device=dev)
        ops.corr_fwd(backend.lib, ops.view(Lt), ops.view(Rt), ops.view(out), md, 1)
        backend.sync()
        assert np.abs(out.cpu().numpy() - gold).max() <= ATOL
    # literal launcher signature: W-padded NHWC in, NCHW out (Nets/Native/shift_corr.cc:22-23)
    Lp_, Rp_ = _padded(L, md, dev), _padded(R, md, dev)
    o2 = torch.full((B, D, H, W), float("nan"), device=dev)
    backend.lib.shift_corr(ops._p(Lp_), ops._p(Rp_), md, B, H, W + 2 * md, Cc, ops._p(o2), None)
    backend.sync()
    assert np.abs(o2.cpu().numpy().transpose(0, 2, 3, 1) - gold).max() <= ATOL


@pytest.mark.parametrize("case", [("g1", 2, 4, 19, 32, 2), ("g2", 1, 3, 11, 6, 3), ("g3", 1, 2, 24, 64, 10)])
def test_shift_corr_grad_is_gradient_of_reference_forward(backend, case):
    """mh_shift_corr_grad (the literal ShiftCorrGradKernelLauncher signature, shift_corr.cc:58-60) = autograd of the reference
    forward formula w.r.t. the PADDED inputs, NHWC."""
    name, B, H, W, Cc, md = case
    dev = backend.device
    rng = np.random.default_rng(7)
    D = 2 * md + 1
    L = rng.standard_normal((B, H, W, Cc)).astype(np.float32); R = rng.standard_normal((B, H, W, Cc)).astype(np.float32)
    g = rng.standard_normal((B, D, H, W)).astype(np.float32)
    Lp_ = _padded(L, md, "cpu").double().requires_grad_(True); Rp_ = _padded(R, md, "cpu").double().requires_grad_(True)
    # out[b,d,y,x] = mean_c in0[b,y,x+md,c] * in1[b,y,x+d,c]   (shift_corr.cu.cc:27-66)
    out = torch.stack([(Lp_[:, :, md:md + W] * Rp_[:, :, d:d + W]).mean(-1) for d in range(D)], 1)
    (out * torch.from_numpy(g).double()).sum().backward()
    o0 = torch.full(Lp_.shape, float("nan"), device=dev); o1 = torch.full(Lp_.shape, float("nan"), device=dev)
    a, b, gg = Lp_.detach().float().to(dev), Rp_.detach().float().to(dev), torch.from_numpy(g).to(dev)
    backend.lib.shift_corr_grad(ops._p(a), ops._p(b), ops._p(gg), md, B, H, W + 2 * md, Cc, ops._p(o0), ops._p(o1), None)
    backend.sync()
    assert (o0.cpu().double() - Lp_.grad).abs().max() <= 1e-5
    assert (o1.cpu().double() - Rp_.grad).abs().max() <= 1e-5


# ---- on the MI355X: the reference kernel itself, built for gfx950, at the full shapes ---------------------------------------
FULL = [("madnet_l6", 1, 6, 20, 192, 2), ("madnet_l5", 1, 12, 40, 128, 2), ("madnet_l4", 1, 24, 80, 96, 2),
        ("madnet_l3", 1, 48, 160, 64, 2), ("madnet_l2", 1, 96, 320, 32, 2), ("dispnet", 1, 96, 320, 128, 40)]


@pytest.mark.gpu
@pytest.mark.parametrize("case", [pytest.param(c, id=c[0]) for c in FULL])
def test_reference_kernel_on_gpu_three_way(hip, case):
    so = os.path.join(ROOT, "oracle", "_ref", "libshift_corr_ref.so")
    assert os.path.exists(so), "oracle/_ref/libshift_corr_ref.so must ship with the snapshot (make -C oracle in the build container)"
    ref = C.CDLL(so)
    name, B, H, W, Cc, md = case
    L, R = G.make_inputs(*case)
    D = 2 * md + 1
    Lp_, Rp_ = _padded(L, md, "cuda"), _padded(R, md, "cuda")
    out_ref = torch.full((B, D, H, W), float("nan"), device="cuda")
    torch.cuda.synchronize()
    rc = ref.ref_shift_corr(ops._p(Lp_), ops._p(Rp_), md, B, H, W + 2 * md, Cc, ops._p(out_ref))     # NULL stream + device sync inside
    assert rc == 0
    r = out_ref.permute(0, 2, 3, 1).cpu()
    assert torch.isfinite(r).all()
    # (a) reproduces the committed fixture where the shapes overlap (rows are independent: the fixture is a crop in H / W)
    if name in ("madnet_l6", "madnet_l5", "madnet_l4") :
        assert np.abs(r.numpy() - GOLD[name + "/out"].transpose(0, 2, 3, 1)).max() <= ATOL
    # (b) the product kernels
    Lt, Rt = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.full((B, H, W, D), float("nan"), device="cuda")
    ops.corr_fwd(hip.lib, ops.view(Lt), ops.view(Rt), ops.view(out), md, 1)
    o2 = torch.full((B, D, H, W), float("nan"), device="cuda")
    hip.lib.shift_corr(ops._p(Lp_), ops._p(Rp_), md, B, H, W + 2 * md, Cc, ops._p(o2), None)
    torch.cuda.synchronize()
    assert (out.cpu() - r).abs().max().item() <= ATOL
    assert (o2.permute(0, 2, 3, 1).cpu() - r).abs().max().item() <= ATOL
    # (c) the oracle
    assert (T.correlation(torch.from_numpy(L), torch.from_numpy(R), md, 1) - r).abs().max().item() <= ATOL
