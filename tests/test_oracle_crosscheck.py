"""The oracle is 'parity unpinned' (no TF, no reference tests), so it is pinned against ITSELF three
ways: torch restatement (oracle/tf_ops.py) vs loop-level numpy (oracle/loops.py) vs plain C
(oracle/c/oracle_ops.c), plus fp64 finite differences of the custom gradient conventions, plus the
closed-form integer paths of SURVEY App. A (pad offsets 4/5/19/19, crop offsets (4,19), SAME pads)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import loops as Lp
from oracle import tf_ops as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def clib():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    return C.CDLL(os.path.join(ROOT, "oracle", "c", "liboracle_ops.so"))


def _r(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def test_integer_paths():
    # even input, stride 2: k=3 -> (0,1); k=5 -> (1,2); k=7 -> (2,3); stride 1 symmetric
    assert T.same_pad(384, 3, 2) == (192, 0, 1) and Lp.same_pad(384, 3, 2) == (192, 0, 1)
    assert T.same_pad(384, 5, 2)[1:] == (1, 2) and T.same_pad(384, 7, 2)[1:] == (2, 3)
    assert T.same_pad(96, 3, 1, 16) == (96, 16, 16)
    # preprocessing.pad_image at 375x1242 -> 384x1280 with 4/5/19/19 ; crop offsets (4,19) undo it
    x = torch.arange(375 * 1242, dtype=torch.float32).reshape(1, 375, 1242, 1)
    p = T.pad_image(x, 64)
    assert p.shape == (1, 384, 1280, 1)
    assert torch.equal(T.center_crop(p, 375, 1242), x)
    assert p[0, 0, 19, 0] == x[0, 4, 0, 0] and p[0, 383, 19, 0] == x[0, 369, 0, 0]   # reflect, edge excluded
    assert p[0, 4, 0, 0] == x[0, 0, 19, 0] and p[0, 4, 1279, 0] == x[0, 0, 1222, 0]
    # legacy resize indices: x2 and x64 up-scaling, 384 -> 375 is not used (crop instead)
    lo, hi, t = Lp.resize_indices(12, 6)
    assert lo.tolist() == [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5] and hi[-1] == 5 and t[1] == 0.5
    # PNG encoding of the disparity dump
    assert Lp.disparity_png(np.array([-1.0, 0.5, 255.99])).tolist() == [0, 128, 65533]


def test_pad_image_loops_vs_torch():
    x = _r((2, 11, 14, 3), 1)
    assert np.array_equal(Lp.pad_image(x, 8), T.pad_image(torch.from_numpy(x), 8).numpy())


@pytest.mark.parametrize("case", [(7, 9, 5, 6, 3, 1, 1), (8, 10, 4, 5, 3, 2, 1), (9, 7, 3, 4, 3, 1, 3), (8, 8, 3, 2, 5, 2, 1), (6, 9, 2, 3, 7, 2, 1)])
def test_conv_three_way(clib, case):
    H, W, Ci, Co, k, s, d = case
    x, w, b = _r((2, H, W, Ci), 2), _r((k, k, Ci, Co), 3, 0.3), _r((Co,), 4)
    ref = T.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), s, d, 0.2).numpy()
    lp = Lp.conv2d(x, w, b, s, d, 0.2)
    assert np.abs(lp - ref).max() < 1e-9
    Ho, Wo = ref.shape[1], ref.shape[2]
    y = np.zeros((2, Ho, Wo, Co), np.float32)
    fp = lambda a: a.ctypes.data_as(C.c_void_p)
    clib.oc_conv2d(fp(x), fp(w), fp(b), fp(y), 2, H, W, Ci, Co, k, k, s, d, C.c_float(0.2))
    assert np.abs(y - ref).max() < 1e-5


@pytest.mark.parametrize("case", [(3, 9, 8, 2, 1), (2, 11, 4, 3, 2), (2, 30, 8, 10, 1)])
def test_corr_three_way_and_grad(clib, case):
    H, W, Cc, md, st = case
    L, R = _r((2, H, W, Cc), 5), _r((2, H, W, Cc), 6)
    Lt = torch.from_numpy(L).double().requires_grad_(True); Rt = torch.from_numpy(R).double().requires_grad_(True)
    ref = T.correlation(Lt, Rt, md, st)
    assert np.abs(Lp.correlation(L, R, md, st) - ref.detach().numpy()).max() < 1e-12
    D = ref.shape[-1]
    out = np.zeros((2, H, W, D), np.float32)
    fp = lambda a: a.ctypes.data_as(C.c_void_p)
    clib.oc_corr_fwd(fp(L), fp(R), fp(out), 2, H, W, Cc, md, st)
    assert np.abs(out - ref.detach().numpy()).max() < 1e-6
    g = _r((2, H, W, D), 7)
    gl, gr = torch.autograd.grad(ref, [Lt, Rt], torch.from_numpy(g).double())
    dL = np.zeros_like(L); dR = np.zeros_like(R)
    clib.oc_corr_bwd(fp(L), fp(R), fp(g), fp(dL), fp(dR), 2, H, W, Cc, md, st)      # SURVEY A.14 closed form
    assert np.abs(dL - gl.numpy()).max() < 1e-5 and np.abs(dR - gr.numpy()).max() < 1e-5


@pytest.mark.parametrize("case", [(6, 20, 12, 40), (5, 7, 13, 9), (9, 11, 5, 7), (7, 7, 7, 7)])
def test_resize_three_way(clib, case):
    H, W, oh, ow = case
    x = _r((2, H, W, 3), 8)
    ref = T.resize_bilinear(torch.from_numpy(x), oh, ow).numpy()
    assert np.abs(Lp.resize_bilinear(x, oh, ow, np.float32) - ref).max() < 1e-6
    y = np.zeros((2, oh, ow, 3), np.float32)
    fp = lambda a: a.ctypes.data_as(C.c_void_p)
    clib.oc_resize(fp(x), fp(y), 2, H, W, 3, oh, ow)
    assert np.abs(y - ref).max() < 1e-6


def test_warpers_loops_vs_torch():
    img = _r((1, 5, 9, 4), 9); u = _r((1, 5, 9, 1), 10, 3.0)
    u[0, 0, 0, 0] = -2.0; u[0, 1, 8, 0] = 4.0; u[0, 2, 3, 0] = 1.0
    a = T.linear_warp(torch.from_numpy(img), torch.from_numpy(u)).numpy()
    assert np.abs(Lp.linear_warp(img, u, np.float32) - a).max() < 1e-6
    b = T.warp_image(torch.from_numpy(img), torch.from_numpy(u)).numpy()
    assert np.abs(Lp.warp_image(img, u, np.float32) - b).max() < 1e-6
    # out-of-range taps: linear_warp -> zero weight; warp_image -> border replicate (App. D.7)
    big = np.full((1, 5, 9, 1), 100.0, np.float32)
    assert np.all(T.linear_warp(torch.from_numpy(img), torch.from_numpy(big)).numpy() == 0)
    assert np.allclose(T.warp_image(torch.from_numpy(img), torch.from_numpy(big)).numpy(), img[:, :, :1, :])


def test_loss_loops_vs_torch():
    x = np.random.default_rng(11).random((1, 7, 9, 3)).astype(np.float32)
    y = np.random.default_rng(12).random((1, 7, 9, 3)).astype(np.float32)
    a = T.mean_ssim_l1(torch.from_numpy(x).double(), torch.from_numpy(y).double()).item()
    assert abs(Lp.mean_ssim_l1(x, y) - a) < 1e-12


def test_gradient_conventions_fp64():
    """leaky'(0)=alpha, relu'(0)=0, |x|'(0)=0, floor'=0, gather gives no index gradient (A.7/A.8)."""
    x = torch.tensor([-1.0, 0.0, 2.0], dtype=torch.float64, requires_grad=True)
    (g,) = torch.autograd.grad(T.leaky(x, 0.2).sum(), x)
    assert g.tolist() == [0.2, 0.2, 1.0]
    (g,) = torch.autograd.grad(torch.relu(x).sum(), x)
    assert g.tolist() == [0.0, 0.0, 1.0]
    # finite differences of the full loss w.r.t. the disparity (away from the floor() kinks)
    gen = torch.Generator().manual_seed(0)
    left = torch.rand(1, 8, 12, 3, generator=gen, dtype=torch.float64) * 255
    right = torch.rand(1, 8, 12, 3, generator=gen, dtype=torch.float64) * 255
    d = (torch.rand(1, 8, 12, 1, generator=gen, dtype=torch.float64) * 3 + 0.3).requires_grad_(True)
    loss = T.reprojection_loss(d, left, right)
    (g,) = torch.autograd.grad(loss, d)
    eps = 1e-6
    for (yy, xx) in [(2, 5), (4, 7), (6, 3)]:
        dp = d.detach().clone(); dp[0, yy, xx, 0] += eps
        dm = d.detach().clone(); dm[0, yy, xx, 0] -= eps
        fd = (T.reprojection_loss(dp, left, right) - T.reprojection_loss(dm, left, right)) / (2 * eps)
        assert abs(fd.item() - g[0, yy, xx, 0].item()) < 1e-6 * max(1.0, abs(fd.item()))


def test_conv_transpose_is_conv_gradient():
    """tf.nn.conv2d_transpose == input-gradient of the SAME conv."""
    x = torch.randn(1, 5, 6, 4, dtype=torch.float64)
    w = torch.randn(4, 4, 7, 4, dtype=torch.float64)          # [kh,kw,Cout,Cin]
    y = T.conv2d_transpose(x, w, torch.zeros(7, dtype=torch.float64), 2, 1.0)
    z = torch.zeros(1, 10, 12, 7, dtype=torch.float64, requires_grad=True)
    f = T.conv2d(z, w, torch.zeros(4, dtype=torch.float64), stride=2)        # HWIO = [4,4,7,4]
    (gz,) = torch.autograd.grad(f, z, x)
    assert torch.allclose(y, gz, atol=1e-12)
