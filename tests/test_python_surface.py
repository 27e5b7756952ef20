"""The reference-named Python surface either side of the kernels: Data_utils.preprocessing (pad_image, rescale_image,
resize_to_prediction, bilinear_sampler, warp_image) and Losses.loss_factory.get_reprojection_loss -- values AND autograd
gradients against the oracle restatement of Data_utils/preprocessing.py:7-29,121-230,269-277 and
Losses/loss_factory.py:353-395.  Runs on the CPU emulator build (the wrappers' `_lib` hook is pointed at it -- test plumbing;
the product resolves `_lib()` to the HIP library and fails without a GPU) and, marked gpu, on the MI355X."""
import numpy as np
import pytest
import torch

from oracle import tf_ops as T


@pytest.fixture
def surface(backend, monkeypatch):
    from Data_utils import preprocessing as P
    from Losses import loss_factory as LF
    monkeypatch.setattr(P, "_lib", lambda: backend.lib)
    monkeypatch.setattr(LF, "_lib", lambda: backend.lib)
    return P, LF, backend


def _leaf(t, dev="cpu"):
    return t.detach().clone().to(dev).requires_grad_(True)


def _r(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def test_pad_image(surface):
    P, _, be = surface
    x = _r((2, 11, 14, 3), 1)
    out = P.pad_image(x.to(be.device), 8)
    be.sync()
    assert torch.equal(out.cpu(), T.pad_image(x, 8))
    # the two networks' call: 375x1242 -> 384x1280 with 4/5/19/19 (preprocessing.py:7-29 with down_factor 64)
    y = _r((1, 375, 1242, 3), 2)
    assert torch.equal(P.pad_image(y.to(be.device), 64).cpu(), T.pad_image(y, 64))


@pytest.mark.parametrize("case", [(2, 7, 9, 1, 14, 18), (1, 12, 20, 3, 6, 10), (1, 9, 11, 3, 4, 5), (1, 6, 8, 2, 6, 8)])
def test_rescale_image_and_gradient(surface, case):
    """tf.image.resize_images legacy bilinear, any channel count (scale_tensor on the RGB frames, Stereo_Online_Adaptation.py:22-23),
    up and down; gradient vs autograd of the oracle."""
    P, _, be = surface
    B, H, W, C, oh, ow = case
    x = _r((B, H, W, C), 3)
    xd = _leaf(x, be.device)
    out = P.rescale_image(xd, [oh, ow])
    xc = _leaf(x)
    ref = T.resize_bilinear(xc, oh, ow)
    if (H, W) == (oh, ow):
        assert out is xd                                          # identity, like the oracle / TF at equal size
        return
    g = _r(ref.shape, 4)
    out.backward(g.to(be.device))
    ref.backward(g)
    be.sync()
    assert (out.detach().cpu() - ref.detach()).abs().max().item() <= 1e-5
    assert (xd.grad.cpu() - xc.grad).abs().max().item() <= 1e-5
    assert P.resize_to_prediction(xd, torch.empty(B, oh, ow, 1)).shape == (B, oh, ow, C)


def test_bilinear_sampler_general(surface):
    """arbitrary coordinates incl. far outside the image (border clamp, un-masked weights -- App. D.7), different source and
    target sizes; gradients w.r.t. coordinates and image."""
    P, _, be = surface
    B, Hs, Ws, C, Ht, Wt = 2, 7, 9, 3, 5, 6
    img = _r((B, Hs, Ws, C), 5)
    rng = np.random.default_rng(6)
    coords = torch.from_numpy(np.stack([rng.uniform(-3, Ws + 2, (B, Ht, Wt)), rng.uniform(-2, Hs + 1, (B, Ht, Wt))], -1).astype(np.float32))
    imd = _leaf(img, be.device); cd = _leaf(coords, be.device)
    out = P.bilinear_sampler(imd, cd)
    ic = _leaf(img); cc = _leaf(coords)
    ref = T.bilinear_sampler(ic, cc)
    g = _r(ref.shape, 7)
    out.backward(g.to(be.device)); ref.backward(g)
    be.sync()
    assert (out.detach().cpu() - ref.detach()).abs().max().item() <= 1e-5
    assert (cd.grad.cpu() - cc.grad).abs().max().item() <= 1e-4
    assert (imd.grad.cpu() - ic.grad).abs().max().item() <= 1e-5


def test_bilinear_sampler_non_finite_coordinates_stay_in_bounds(surface):
    """NaN / Inf / huge coordinates (a diverged network feeding preprocessing.bilinear_sampler): the tap indices are clamped BEFORE the integer cast
    (fmaxf / fminf return the non-NaN operand), so the gather and the gradient scatter stay inside the image; the pixels with finite coordinates
    are what they are without the bad ones, and huge finite coordinates follow the oracle (weights 0 in fp32)."""
    P, _, be = surface
    B, Hs, Ws, C, Ht, Wt = 1, 6, 8, 2, 4, 5
    img = _r((B, Hs, Ws, C), 21)
    rng = np.random.default_rng(22)
    good = torch.from_numpy(np.stack([rng.uniform(-1, Ws, (B, Ht, Wt)), rng.uniform(-1, Hs, (B, Ht, Wt))], -1).astype(np.float32))
    bad = good.clone()
    bad[0, 0, 0, 0] = float("nan"); bad[0, 0, 1, 1] = float("nan"); bad[0, 1, 0, 0] = float("inf"); bad[0, 1, 1, 1] = float("-inf")
    bad[0, 2, 0] = torch.tensor([1e30, -1e30]); bad[0, 2, 1] = torch.tensor([-3e9, 3e9])
    finite = torch.ones(B, Ht, Wt, dtype=torch.bool); finite[0, 0, :2] = False; finite[0, 1, :2] = False
    outs = {}
    for name, c in (("good", good), ("bad", bad)):
        imd = _leaf(img, be.device); cd = _leaf(c, be.device)
        out = P.bilinear_sampler(imd, cd)
        out.backward(torch.ones_like(out))
        be.sync()
        outs[name] = (out.detach().cpu(), cd.grad.cpu(), imd.grad.cpu())
    ref = T.bilinear_sampler(img, torch.where(finite[..., None], bad, good))       # (the oracle, like TF's gather, rejects NaN indices)
    keep = finite.clone(); keep[0, 2, :2] = False                      # pixels whose coordinates are the same in both runs
    assert torch.equal(outs["bad"][0][keep], outs["good"][0][keep]) and torch.equal(outs["bad"][1][keep], outs["good"][1][keep])
    assert (outs["bad"][0][finite] - ref[finite]).abs().max().item() <= 1e-5          # incl. the 1e30 / 3e9 ones
    assert torch.isfinite(outs["bad"][0][finite]).all()


def test_warp_image_matches_oracle_and_fused_loss_kernel(surface):
    """preprocessing.warp_image = bilinear_sampler at (x - d, y): equals the oracle's warp_image, including disparities that
    leave the image; its gradient w.r.t. the disparity is what the fused loss kernel differentiates."""
    P, _, be = surface
    B, H, W = 1, 9, 21
    img = (torch.rand(B, H, W, 3, generator=torch.Generator().manual_seed(8)) * 255).floor()
    d = torch.rand(B, H, W, 1, generator=torch.Generator().manual_seed(9)) * 30 - 5
    dd = _leaf(d, be.device)
    out = P.warp_image(img.to(be.device), dd)
    dc = _leaf(d)
    ref = T.warp_image(img, dc)
    g = _r(ref.shape, 10)
    out.backward(g.to(be.device)); ref.backward(g)
    be.sync()
    assert (out.detach().cpu() - ref.detach()).abs().max().item() <= 1e-3 * 1e-2 * 255        # |img| <= 255
    assert (dd.grad.cpu() - dc.grad).abs().max().item() <= 1e-3 * max(1.0, dc.grad.abs().max().item())


@pytest.mark.parametrize("size", [(24, 33), (375 // 5, 1242 // 6)])
def test_get_reprojection_loss_value_and_gradient(surface, size):
    """Losses.loss_factory.get_reprojection_loss('mean_SSIM_l1', reduced=True)(disparities, inputs) (loss_factory.py:353-395) as
    the online script calls it (Stereo_Online_Adaptation.py:70): value + gradient w.r.t. the last disparity vs the oracle."""
    _, LF, be = surface
    from madnet_hip import synthetic as S
    H, W = size
    l, r, gt = S.make_pair(H, W)
    disp = torch.from_numpy(gt).clone() * 0 + torch.rand(1, H, W, 1, generator=torch.Generator().manual_seed(11)) * 12
    dd = _leaf(disp, be.device)
    inputs = {"left": torch.from_numpy(l).to(be.device), "right": torch.from_numpy(r).to(be.device), "target": torch.from_numpy(gt).to(be.device)}
    loss = LF.get_reprojection_loss("mean_SSIM_l1", reduced=True)([torch.zeros(1, 4, 4, 1, device=be.device), dd], inputs)
    loss.backward()
    dc = _leaf(disp)
    ref = T.reprojection_loss(dc, torch.from_numpy(l), torch.from_numpy(r))
    ref.backward()
    be.sync()
    assert abs(loss.item() - ref.item()) <= 2e-5 * max(1.0, abs(ref.item()))
    assert (dd.grad.cpu() - dc.grad).abs().max().item() <= 2e-3 * dc.grad.abs().max().item()
    with pytest.raises(Exception):
        LF.get_reprojection_loss("no_such_loss")
    # multiScale + a low-resolution prediction: resize_to_prediction * (W_img / W_disp), weighted sum (loss_factory.py:376-392)
    small = torch.rand(1, H // 2, W // 2, 1, generator=torch.Generator().manual_seed(12)) * 6
    acc = LF.get_reprojection_loss("mean_SSIM_l1", multiScale=True, weights=[0.5, 2.0], reduced=False)([small.to(be.device), dd.detach()], inputs)
    be.sync()
    r0 = T.reprojection_loss(disp, torch.from_numpy(l), torch.from_numpy(r)).item()
    r1 = T.reprojection_loss(small, torch.from_numpy(l), torch.from_numpy(r)).item()
    assert abs(acc[0].item() - 0.5 * r0) <= 2e-5 and abs(acc[1].item() - 2.0 * r1) <= 4e-5
