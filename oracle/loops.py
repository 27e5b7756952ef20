"""CPU ORACLE (test infrastructure, NOT product code) -- loop-level numpy restatement.

Independent second restatement of the integer/index paths and of the small-shape
arithmetic of the hot path, written with explicit index loops (no torch, no
library conv / resize / gather), used by tests/ to cross-check oracle/tf_ops.py.
PARITY UNPINNED (no TF here; the reference has no tests) -- see oracle/tf_ops.py.
"""
import numpy as np


def same_pad(in_size, k, stride, dilation=1):
    """out=ceil(in/s); total=max((out-1)*s+keff-in,0); before=total//2."""
    keff = (k - 1) * dilation + 1
    out = (in_size + stride - 1) // stride
    total = max((out - 1) * stride + keff - in_size, 0)
    return out, total // 2, total - total // 2


def reflect_index(i, n):
    """tf.pad(mode='REFLECT') source index for (possibly out-of-range) i."""
    if i < 0:
        return -i
    if i >= n:
        return 2 * (n - 1) - i
    return i


def pad_image(x, down_factor=64):
    """preprocessing.py:7-29."""
    B, H, W, C = x.shape
    nh = H if H % down_factor == 0 else (H // down_factor + 1) * down_factor
    nw = W if W % down_factor == 0 else (W // down_factor + 1) * down_factor
    pt, pl = (nh - H) // 2, (nw - W) // 2
    out = np.empty((B, nh, nw, C), x.dtype)
    for y in range(nh):
        sy = reflect_index(y - pt, H)
        for xx in range(nw):
            out[:, y, xx] = x[:, sy, reflect_index(xx - pl, W)]
    return out


def conv2d(x, w, b, stride=1, dilation=1, alpha=1.0, dtype=np.float64):
    """sharedLayers.py:54-77, tap loops, accumulation in `dtype`."""
    B, H, W, Ci = x.shape
    kh, kw, _, Co = w.shape
    Ho, pt, _ = same_pad(H, kh, stride, dilation)
    Wo, pl, _ = same_pad(W, kw, stride, dilation)
    out = np.zeros((B, Ho, Wo, Co), dtype)
    xw = x.astype(dtype); ww = w.astype(dtype)
    for oy in range(Ho):
        for ox in range(Wo):
            acc = np.zeros((B, Co), dtype)
            for ky in range(kh):
                iy = oy * stride + ky * dilation - pt
                if iy < 0 or iy >= H:
                    continue
                for kx in range(kw):
                    ix = ox * stride + kx * dilation - pl
                    if ix < 0 or ix >= W:
                        continue
                    acc += xw[:, iy, ix, :] @ ww[ky, kx]
            out[:, oy, ox] = acc + b.astype(dtype)
    return np.where(out > 0, out, alpha * out) if alpha != 1.0 else out


def correlation(x, y, max_disp, stride=1, dtype=np.float64):
    """sharedLayers.py:41-51."""
    B, H, W, C = x.shape
    shifts = list(range(-max_disp, max_disp + 1, stride))
    out = np.zeros((B, H, W, len(shifts)), dtype)
    for j, i in enumerate(shifts):
        for xx in range(W):
            xs = xx + i
            if 0 <= xs < W:
                out[:, :, xx, j] = (x[:, :, xx, :].astype(dtype) * y[:, :, xs, :].astype(dtype)).sum(-1) / C
    return out


def resize_indices(out_size, in_size):
    """SURVEY A.4 index path: float32 scale, lower=int(src), upper=min(lower+1,in-1)."""
    scale = np.float32(in_size) / np.float32(out_size)
    lo, hi, t = [], [], []
    for i in range(out_size):
        src = np.float32(np.float32(i) * scale)
        l = int(src)
        lo.append(l); hi.append(min(l + 1, in_size - 1)); t.append(np.float32(src - np.float32(l)))
    return np.array(lo), np.array(hi), np.array(t, np.float32)


def resize_bilinear(x, oh, ow, dtype=np.float64):
    B, H, W, C = x.shape
    if (H, W) == (oh, ow):
        return x.astype(dtype)
    ylo, yhi, ty = resize_indices(oh, H)
    xlo, xhi, tx = resize_indices(ow, W)
    out = np.empty((B, oh, ow, C), dtype)
    xx = x.astype(dtype)
    for y in range(oh):
        for z in range(ow):
            tl, tr = xx[:, ylo[y], xlo[z]], xx[:, ylo[y], xhi[z]]
            bl, br = xx[:, yhi[y], xlo[z]], xx[:, yhi[y], xhi[z]]
            top = tl + (tr - tl) * dtype(tx[z])
            bot = bl + (br - bl) * dtype(tx[z])
            out[:, y, z] = top + (bot - top) * dtype(ty[y])
    return out


def linear_warp(img, u, dtype=np.float64):
    """MadNet.py:378-436 (zero weight outside)."""
    B, H, W, C = img.shape
    out = np.zeros((B, H, W, C), dtype)
    for b in range(B):
        for y in range(H):
            for x in range(W):
                cx = np.float32(np.float32(x) + np.float32(u[b, y, x, 0]))
                x0 = np.floor(cx); x1 = x0 + 1
                x0s = min(max(x0, 0.0), W - 1.0); x1s = min(max(x1, 0.0), W - 1.0)
                w0 = np.float32(x1 - cx) * (1.0 if x0 == x0s else 0.0)
                w1 = np.float32(cx - x0) * (1.0 if x1 == x1s else 0.0)
                out[b, y, x] = dtype(w0) * img[b, y, int(x0s)].astype(dtype) + dtype(w1) * img[b, y, int(x1s)].astype(dtype)
    return out


def warp_image(img, disp, dtype=np.float64):
    """preprocessing.py:121-230 with all four taps, clamped indices, unmasked weights."""
    B, H, W, C = img.shape
    out = np.zeros((B, H, W, C), dtype)
    flat = img.reshape(B, H * W, C).astype(dtype)
    for b in range(B):
        for y in range(H):
            for x in range(W):
                cx = np.float32(np.float32(x) - np.float32(disp[b, y, x, 0])); cy = np.float32(y)
                x0 = np.floor(cx); x1 = x0 + 1; y0 = np.floor(cy); y1 = y0 + 1
                wx0 = np.float32(x1 - cx); wx1 = np.float32(cx - x0)
                wy0 = np.float32(y1 - cy); wy1 = np.float32(cy - y0)
                cl = lambda v, m: min(max(v, 0.0), m)
                x0s, x1s = cl(x0, W - 1.0), cl(x1, W - 1.0)
                y0s, y1s = cl(y0, H - 1.0), cl(y1, H - 1.0)
                idx = lambda ys, xs: int(np.float32(np.float32(ys) * np.float32(W)) + np.float32(xs))
                out[b, y, x] = (dtype(wx0 * wy0) * flat[b, idx(y0s, x0s)] + dtype(wx0 * wy1) * flat[b, idx(y1s, x0s)] +
                                dtype(wx1 * wy0) * flat[b, idx(y0s, x1s)] + dtype(wx1 * wy1) * flat[b, idx(y1s, x1s)])
    return out


def mean_ssim_l1(x, y, dtype=np.float64):
    """loss_factory.py:128-164."""
    B, H, W, C = x.shape
    x = x.astype(dtype); y = y.astype(dtype)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    tot = 0.0
    for b in range(B):
        for yy in range(H - 2):
            for xx in range(W - 2):
                px = x[b, yy:yy + 3, xx:xx + 3].reshape(9, C); py = y[b, yy:yy + 3, xx:xx + 3].reshape(9, C)
                mx, my = px.mean(0), py.mean(0)
                sx = (px ** 2).mean(0) - mx ** 2; sy = (py ** 2).mean(0) - my ** 2
                sxy = (px * py).mean(0) - mx * my
                n = (2 * mx * my + C1) * (2 * sxy + C2); d = (mx ** 2 + my ** 2 + C1) * (sx + sy + C2)
                tot += np.clip((1 - n / d) / 2, 0, 1).sum()
    ssim = tot / (B * (H - 2) * (W - 2) * C)
    return 0.85 * ssim + 0.15 * np.abs(x - y).mean()


def disparity_png(d):
    """Stereo_Online_Adaptation.py:246-251: (clip(d,0,256)*256).astype(uint16)."""
    return (np.clip(d, 0, 256) * 256.0).astype(np.uint16)
