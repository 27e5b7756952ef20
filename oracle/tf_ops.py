"""CPU ORACLE (test infrastructure, NOT product code) -- TF 1.12 op semantics on torch-CPU.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

PARITY: the reference ships no tests / golden vectors and TensorFlow 1.x cannot be imported in this
environment, so the functions here restate the *documented* TF 1.12 kernel semantics
-- "parity unpinned" for the arithmetic INSIDE TensorFlow's library kernels (conv2d, resize_images,
avg_pool, ...).  They are cross-checked against an independent loop-level numpy restatement
(oracle/loops.py), a plain-C restatement (oracle/c/oracle_ops.c) and fp64 finite differences (tests/).
What IS pinned to the reference itself: the correlation (the reference's own kernel, oracle/_ref), the
samplers (the reference module imported) and -- round 4 -- the whole graph WIRING around these kernels:
oracle/tf_shim binds the reference's `tf.*` calls to the functions below and oracle/ref_graph.py EXECUTES
/root/reference/Nets/*.py, Losses/loss_factory.py, Data_utils/preprocessing.py as written
(tests/test_ref_graph.py, fixtures tests/golden/ref_graph_*.npz).

All tensors at this API are NHWC like the reference; torch kernels are called NCHW
internally.  Each function cites the reference call-site it restates.
"""
import math
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# activations
# ----------------------------------------------------------------------------
def leaky(x, alpha):
    """tf.maximum(alpha*x, x)  (Nets/MadNet.py:366-367, Nets/sharedLayers.py:54).

    TF MaximumGrad routes the gradient to the first argument when alpha*x >= x,
    i.e. d/dx = alpha for x <= 0 INCLUDING x == 0, 1 for x > 0.
    """
    if alpha == 1.0:
        return x
    return torch.where(x > 0, x, alpha * x)


# ----------------------------------------------------------------------------
# padding helpers
# ----------------------------------------------------------------------------
def same_pad(in_size, k, stride, dilation=1):
    """TF 'SAME' padding: returns (out, pad_before, pad_after)."""
    keff = (k - 1) * dilation + 1
    out = -(-in_size // stride)
    total = max((out - 1) * stride + keff - in_size, 0)
    before = total // 2
    return out, before, total - before


def pad_image(x, down_factor=64):
    """Data_utils/preprocessing.py:7-29 (static branch): REFLECT pad H,W up to a
    multiple of down_factor; before=(new-old)//2, after=(new-old+1)//2."""
    h, w = x.shape[1], x.shape[2]
    nh = h if h % down_factor == 0 else (h // down_factor + 1) * down_factor
    nw = w if w % down_factor == 0 else (w // down_factor + 1) * down_factor
    pt, pb = (nh - h) // 2, (nh - h + 1) // 2
    pl, pr = (nw - w) // 2, (nw - w + 1) // 2
    if pt == pb == pl == pr == 0:
        return x
    xn = x.permute(0, 3, 1, 2)
    xn = F.pad(xn, (pl, pr, pt, pb), mode="reflect")
    return xn.permute(0, 2, 3, 1)


def center_crop(x, th, tw):
    """tf.image.resize_image_with_crop_or_pad, crop branch:
    offset = (in - target)//2  (Nets/MadNet.py:70,363)."""
    h, w = x.shape[1], x.shape[2]
    oy, ox = (h - th) // 2, (w - tw) // 2
    return x[:, oy:oy + th, ox:ox + tw, :]


# ----------------------------------------------------------------------------
# convolutions (Nets/sharedLayers.py:54-92)
# ----------------------------------------------------------------------------
def conv2d(x, w_hwio, b, stride=1, dilation=1, alpha=1.0):
    """sharedLayers.conv2d / dilated_conv2d: tf.nn.conv2d(NHWC,HWIO,'SAME') (or
    atrous_conv2d) + bias_add + leaky(alpha) (alpha=1 -> linear).
    Cross-correlation, asymmetric SAME pad for stride 2."""
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    _, pt, pb = same_pad(x.shape[1], kh, stride, dilation)
    _, pl, pr = same_pad(x.shape[2], kw, stride, dilation)
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    wn = w_hwio.permute(3, 2, 0, 1).contiguous()
    y = F.conv2d(xn, wn, b, stride=stride, dilation=dilation)
    return leaky(y.permute(0, 2, 3, 1), alpha)


def conv2d_transpose(x, w_hwoi, b, stride=2, alpha=1.0):
    """sharedLayers.conv2d_transpose (sharedLayers.py:80-92): tf.nn.conv2d_transpose
    with W[kh,kw,Cout,Cin], output = in*stride, 'SAME' == gradient of a SAME conv
    (SURVEY A.3).  Restated literally as that gradient: dgrad of conv2d(k,stride)."""
    kh, kw, cout, cin = w_hwoi.shape
    B, H, W, _ = x.shape
    Ho, Wo = H * stride, W * stride
    _, pt, _ = same_pad(Ho, kh, stride)
    _, pl, _ = same_pad(Wo, kw, stride)
    # forward conv that this op is the gradient of: in [B,Ho,Wo,cout] -> [B,H,W,cin]
    wn = w_hwoi.permute(3, 2, 0, 1).contiguous()  # [cin(out of fwd conv), cout(in of fwd conv), kh, kw]
    xn = x.permute(0, 3, 1, 2)
    # conv_transpose2d(padding=p) crops p from each side of the full output;
    # full size = (H-1)*s + k ; we need offset pt/pl at top/left and size Ho/Wo.
    full = F.conv_transpose2d(xn, wn, None, stride=stride)
    y = full[:, :, pt:pt + Ho, pl:pl + Wo]
    if y.shape[2] < Ho or y.shape[3] < Wo:
        y = F.pad(y, (0, Wo - y.shape[3], 0, Ho - y.shape[2]))
    y = y + b.view(1, -1, 1, 1)
    return leaky(y.permute(0, 2, 3, 1), alpha)


# ----------------------------------------------------------------------------
# correlation (Nets/sharedLayers.py:41-51, the 'TF' formulation is the oracle)
# ----------------------------------------------------------------------------
def correlation(x, y, max_disp, stride=1):
    """corr[b,h,w,j] = mean_c( x[b,h,w,c] * ypad[b,h,w+i+max_disp,c] ),
    i = -max_disp .. max_disp step stride, y zero padded by max_disp in W."""
    W = y.shape[2]
    yp = F.pad(y, (0, 0, max_disp, max_disp))
    outs = []
    for i in range(-max_disp, max_disp + 1, stride):
        shifted = yp[:, :, i + max_disp:i + max_disp + W, :]
        outs.append((shifted * x).mean(dim=-1, keepdim=True))
    return torch.cat(outs, dim=-1)


# ----------------------------------------------------------------------------
# TF1 legacy bilinear resize
# ----------------------------------------------------------------------------
def _interp_weights(out_size, in_size, dtype):
    import numpy as np
    scale = np.float32(in_size) / np.float32(out_size)
    src = (np.arange(out_size, dtype=np.float32) * scale).astype(np.float32)
    lo = src.astype(np.int64)
    hi = np.minimum(lo + 1, in_size - 1)
    lerp = (src - lo.astype(np.float32)).astype(np.float32)
    return torch.from_numpy(lo), torch.from_numpy(hi), torch.from_numpy(lerp).to(dtype)


def resize_bilinear(x, out_h, out_w):
    """tf.image.resize_images(bilinear, align_corners=False), legacy kernel without
    half-pixel centres (Nets/MadNet.py:69,274,362; preprocessing.py:273).
    Identity when the size is unchanged."""
    B, H, W, C = x.shape
    if H == out_h and W == out_w:
        return x
    ylo, yhi, ty = _interp_weights(out_h, H, x.dtype)
    xlo, xhi, tx = _interp_weights(out_w, W, x.dtype)
    top_rows = x.index_select(1, ylo)
    bot_rows = x.index_select(1, yhi)
    tx = tx.view(1, 1, -1, 1)
    ty = ty.view(1, -1, 1, 1)
    tl, tr = top_rows.index_select(2, xlo), top_rows.index_select(2, xhi)
    bl, br = bot_rows.index_select(2, xlo), bot_rows.index_select(2, xhi)
    top = tl + (tr - tl) * tx
    bot = bl + (br - bl) * tx
    return top + (bot - top) * ty


# ----------------------------------------------------------------------------
# warpers
# ----------------------------------------------------------------------------
def linear_warp(imgs, u):
    """MadNet._build_indeces + _linear_warping (Nets/MadNet.py:378-436) with
    coords = [b, x + u, y + 0]: 1-D linear interpolation along x, taps that fall
    outside [0,W-1] get weight ZERO (mask x0==clip(x0)); rows floor(y) clamped."""
    B, H, W, C = imgs.shape
    cdt = u.dtype                       # float32 in the reference graph
    xs = torch.arange(W, dtype=cdt).view(1, 1, W, 1)
    cx = xs + u
    x0 = torch.floor(cx)
    x1 = x0 + 1
    x_max = float(W - 1)
    x0s = torch.clamp(x0, 0.0, x_max)
    x1s = torch.clamp(x1, 0.0, x_max)
    w0 = (x1 - cx) * (x0 == x0s).to(cdt)
    w1 = (cx - x0) * (x1 == x1s).to(cdt)
    i0 = x0s.detach().to(torch.int64).expand(B, H, W, C)
    i1 = x1s.detach().to(torch.int64).expand(B, H, W, C)
    im0 = torch.gather(imgs, 2, i0)
    im1 = torch.gather(imgs, 2, i1)
    return w0.to(imgs.dtype) * im0 + w1.to(imgs.dtype) * im1


def bilinear_sampler(imgs, coords):
    """preprocessing.bilinear_sampler, general form (Data_utils/preprocessing.py:121-199): coords[...,0] = x, [...,1] = y;
    indices clamped to the border, weights un-masked, flat gather index built in the coords' float dtype (float32 in the
    reference graph, :170-187) then cast.  Gradients flow to coords through the weights only (tf.floor has none) and to imgs."""
    B, Hs, Ws, C = imgs.shape
    _, Ht, Wt, _ = coords.shape
    cx, cy = coords[..., 0:1], coords[..., 1:2]
    x0 = torch.floor(cx); x1 = x0 + 1
    y0 = torch.floor(cy); y1 = y0 + 1
    wx0 = x1 - cx; wx1 = cx - x0
    wy0 = y1 - cy; wy1 = cy - y0
    x0s = torch.clamp(x0, 0.0, float(Ws - 1)); x1s = torch.clamp(x1, 0.0, float(Ws - 1))
    y0s = torch.clamp(y0, 0.0, float(Hs - 1)); y1s = torch.clamp(y1, 0.0, float(Hs - 1))
    base = (torch.arange(B, dtype=coords.dtype) * float(Ws * Hs)).view(B, 1, 1, 1)
    flat = imgs.reshape(B * Hs * Ws, C)

    def g(ysafe, xsafe):
        idx = (xsafe + (base + ysafe * float(Ws))).detach().to(torch.int64).reshape(-1)
        return flat.index_select(0, idx).reshape(B, Ht, Wt, C)

    im00, im01, im10, im11 = g(y0s, x0s), g(y1s, x0s), g(y0s, x1s), g(y1s, x1s)
    dt = imgs.dtype
    return ((wx0 * wy0).to(dt) * im00 + (wx0 * wy1).to(dt) * im01 + (wx1 * wy0).to(dt) * im10) + (wx1 * wy1).to(dt) * im11


def warp_image(img, disp):
    """preprocessing.warp_image + bilinear_sampler (preprocessing.py:121-230) with
    coords = (x - d, y).  Indices are clamped to the border with UN-masked weights
    (border replicate).  y is integral so the y1 taps have weight 0."""
    B, H, W, C = img.shape
    cdt = disp.dtype                    # float32 in the reference graph
    xs = torch.arange(W, dtype=cdt).view(1, 1, W, 1)
    ys = torch.arange(H, dtype=cdt).view(1, H, 1, 1)
    cx = xs - disp
    cy = ys.expand(B, H, W, 1)
    x0 = torch.floor(cx); x1 = x0 + 1
    y0 = torch.floor(cy); y1 = y0 + 1
    wx0 = x1 - cx; wx1 = cx - x0
    wy0 = y1 - cy; wy1 = cy - y0
    x0s = torch.clamp(x0, 0.0, float(W - 1)); x1s = torch.clamp(x1, 0.0, float(W - 1))
    y0s = torch.clamp(y0, 0.0, float(H - 1)); y1s = torch.clamp(y1, 0.0, float(H - 1))
    flat = img.reshape(B, H * W, C)

    def g(ysafe, xsafe):
        idx = (ysafe * float(W) + xsafe).detach().to(torch.int64).reshape(B, H * W, 1)
        return torch.gather(flat, 1, idx.expand(B, H * W, C)).reshape(B, H, W, C)

    im00, im01 = g(y0s, x0s), g(y1s, x0s)
    im10, im11 = g(y0s, x1s), g(y1s, x1s)
    dt = img.dtype
    return ((wx0 * wy0).to(dt) * im00 + (wx0 * wy1).to(dt) * im01 +
            (wx1 * wy0).to(dt) * im10 + (wx1 * wy1).to(dt) * im11)


# ----------------------------------------------------------------------------
# loss (Losses/loss_factory.py)
# ----------------------------------------------------------------------------
def _avg_pool3_valid(x):
    return F.avg_pool2d(x.permute(0, 3, 1, 2), 3, 1).permute(0, 2, 3, 1)


def ssim_map(x, y):
    """loss_factory.SSIM (loss_factory.py:128-149)."""
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    mu_x, mu_y = _avg_pool3_valid(x), _avg_pool3_valid(y)
    sigma_x = _avg_pool3_valid(x ** 2) - mu_x ** 2
    sigma_y = _avg_pool3_valid(y ** 2) - mu_y ** 2
    sigma_xy = _avg_pool3_valid(x * y) - mu_x * mu_y
    n = (2 * mu_x * mu_y + C1) * (2 * sigma_xy + C2)
    d = (mu_x ** 2 + mu_y ** 2 + C1) * (sigma_x + sigma_y + C2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def mean_ssim_l1(x, y):
    """loss_factory.mean_SSIM_L1 (loss_factory.py:163-164,156,28-38)."""
    return 0.85 * ssim_map(x, y).mean() + 0.15 * (x - y).abs().mean()


def reprojection_loss(disp, left, right):
    """get_reprojection_loss('mean_SSIM_l1', multiScale=False)([.., disp], inputs)
    (loss_factory.py:353-395): images /256, disparity resized to the image size and
    scaled by W_img/W_disp, right warped to left, mean_SSIM_L1(reprojected, left)."""
    l = left.to(disp.dtype) / 256.0
    r = right.to(disp.dtype) / 256.0
    scale = float(l.shape[2]) / float(disp.shape[2])
    rd = resize_bilinear(disp, l.shape[1], l.shape[2]) * scale
    rep = warp_image(r, rd)
    return mean_ssim_l1(rep, l)


def proxy_loss(pred, proxy, weight=0.01):
    """loss_factory.get_proxy_loss('mean_l1', weights=[weight]*10, reduced=True) on ONE full-resolution prediction
    (Losses/loss_factory.py:304-351; mean_l1 :28-38): valid = !(proxy <= 0 | proxy >= 192) (the literal 192, not max_disp),
    weight * sum(valid*|pred - proxy|) / sum(valid).  resize_to_prediction is the identity at equal sizes."""
    valid = torch.where((proxy <= 0) | (proxy >= 192), torch.zeros_like(proxy), torch.ones_like(proxy))
    return weight * (valid * (pred - proxy).abs()).sum() / valid.sum()


def supervised_loss(pred, target, weight=1.0, max_disp=192.0):
    """ONE scale of loss_factory.get_supervised_loss('mean_l1', multiScale=True, max_disp=MAX_DISP) (Losses/loss_factory.py:256-302,
    Train.py:19-20,100): valid = !(target == 0 | target >= max_disp); weight * sum(valid*|pred - target|) / sum(valid).  Every
    MADNet / DispNet prediction is already full resolution (_make_disp), so resize_to_prediction is the identity and the scale
    factor W_left / W_pred is 1."""
    valid = torch.where((target == 0) | (target >= max_disp), torch.zeros_like(target), torch.ones_like(target))
    return weight * (valid * (pred - target).abs()).sum() / valid.sum()


def adam_update(var, m, v, g, state, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer(lr, beta1).apply_gradients for one variable (TF 1.12 python/training/adam.py + ApplyAdam kernel),
    fp32 throughout: lr_t = lr*sqrt(1 - beta2_power)/(1 - beta1_power); m += (g - m)(1 - b1); v += (g^2 - v)(1 - b2);
    var -= (m * lr_t) / (sqrt(v) + eps).  state = [beta1_power, beta2_power] (advanced by the caller after all variables)."""
    f32 = torch.float32
    one = torch.tensor(1.0, dtype=f32)
    b1p, b2p = torch.tensor(state[0], dtype=f32), torch.tensor(state[1], dtype=f32)
    b1, b2 = torch.tensor(beta1, dtype=f32), torch.tensor(beta2, dtype=f32)
    lr_t = torch.tensor(lr, dtype=f32) * torch.sqrt(one - b2p) / (one - b1p)
    # the ApplyAdam functor (tensorflow/core/kernels/training_ops.cc): m += (g - m)(1 - b1); v += (g^2 - v)(1 - b2)
    m.add_((g - m) * (one - b1))
    v.add_((g * g - v) * (one - b2))
    var.sub_((m * lr_t) / (v.sqrt() + torch.tensor(eps, dtype=f32)))


def validation_metrics(disp, gt, pixel_th=3.0):
    """Stereo_Online_Adaptation.py:74-82: EPE and bad3 over gt != 0."""
    abs_err = (disp - gt).abs()
    valid = (gt != 0).to(disp.dtype)
    filt = abs_err * valid
    nvalid = valid.sum()
    epe = filt.sum() / nvalid
    bad = (filt > pixel_th).to(disp.dtype).sum() / nvalid
    return epe, bad
