"""Synthetic KITTI-shaped stereo pairs and Xavier weights (no dataset / checkpoint is
reachable offline).  Protocol fixed in SURVEY.md 8(d):

* pair generator, seed = 1234 + stream_id: left = multi-octave value-noise texture +
  a few rectangles, quantised to integers 0..255 stored as float32 [1,H,W,3] (the
  reference reader yields float32 of uint8 values, Data_utils/data_reader.py:98);
  disparity smooth in [2,96] px, larger at the bottom; right = left warped by d with
  border clamp, re-quantised; GT = d with ~30% of the pixels kept (0 = invalid).
* weights: Xavier-uniform U(+-sqrt(6/(k*k*Cin + k*k*Cout))), biases 0
  (Nets/sharedLayers.py:4-5), numpy default_rng(seed), generated in manifest order.
"""
import numpy as np


def _value_noise(rng, h, w, octaves=5):
    img = np.zeros((h, w), np.float32)
    amp, tot = 1.0, 0.0
    for o in range(octaves):
        gh, gw = 3 * 2 ** o + 2, 8 * 2 ** o + 2
        g = rng.random((gh, gw), dtype=np.float32)
        ys = np.linspace(0, gh - 1.001, h, dtype=np.float32)
        xs = np.linspace(0, gw - 1.001, w, dtype=np.float32)
        y0 = ys.astype(np.int64); x0 = xs.astype(np.int64)
        ty = (ys - y0)[:, None]; tx = (xs - x0)[None, :]
        a = g[y0][:, x0]; b = g[y0][:, x0 + 1]
        c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
        img += amp * ((a * (1 - tx) + b * tx) * (1 - ty) + (c * (1 - tx) + d * tx) * ty)
        tot += amp
        amp *= 0.6
    return img / tot


def make_pair(h=375, w=1242, stream_id=0, frame=0):
    """Returns left, right [1,h,w,3] float32 (integer values 0..255) and gt [1,h,w,1]."""
    rng = np.random.default_rng(1234 + stream_id)
    wide = w + 256
    tex = np.stack([_value_noise(rng, h, wide) for _ in range(3)], -1)
    for _ in range(6):
        y0 = int(rng.integers(0, h - 20)); x0 = int(rng.integers(0, wide - 40))
        hh = int(rng.integers(10, max(11, h // 4))); ww = int(rng.integers(20, max(21, w // 6)))
        tex[y0:y0 + hh, x0:x0 + ww] = rng.random(3, dtype=np.float32)
    shift = frame % 128
    left = np.floor(tex[:, shift:shift + w] * 255.0 + 0.5).clip(0, 255).astype(np.float32)
    yy = np.linspace(0.0, 1.0, h, dtype=np.float32)[:, None]
    bump = _value_noise(rng, h, w, octaves=2)
    disp = (2.0 + 94.0 * (0.15 + 0.7 * yy) * (0.6 + 0.4 * bump)).astype(np.float32)
    disp = np.clip(disp, 2.0, 96.0)
    xs = np.arange(w, dtype=np.float32)[None, :] + disp      # right(x) = left(x + d) <=> left(x) = right(x - d)
    x0 = np.floor(xs); t = (xs - x0)[..., None]
    i0 = np.clip(x0, 0, w - 1).astype(np.int64); i1 = np.clip(x0 + 1, 0, w - 1).astype(np.int64)
    rows = np.arange(h)[:, None]
    right = left[rows, i0] * (1 - t) + left[rows, i1] * t
    right = np.floor(right + 0.5).clip(0, 255).astype(np.float32)
    # disparity is defined on the LEFT view: left(x) ~= right(x - d_left(x)); use the
    # first-order approximation d_left(x) = disp(x - disp) which is smooth as well.
    xl = np.clip(np.arange(w, dtype=np.float32)[None, :] - disp, 0, w - 1).astype(np.int64)
    d_left = disp[rows, xl]
    keep = rng.random((h, w)) < 0.3
    gt = np.where(keep, d_left, 0.0).astype(np.float32)
    return left[None], right[None], gt[None, :, :, None]


def xavier_weights(shapes, seed=0):
    """shapes: {name: shape}.  Returns {name: float32 ndarray}.  Every tensor has its own stream
    seeded by (seed, crc32(name)), so the values do not depend on the iteration order of `shapes`."""
    import zlib
    out = {}
    for name, shp in shapes.items():
        rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
        if len(shp) == 1:
            out[name] = np.zeros(shp, np.float32)
        else:
            kh, kw, a, b = shp
            lim = np.sqrt(6.0 / (kh * kw * a + kh * kw * b))
            out[name] = rng.uniform(-lim, lim, size=shp).astype(np.float32)
    return out


def calibrated_weights(shapes, seed=1, conv1_gain=0.1, dispnet_pred_gain=60.0):
    """Xavier weights whose first layer is scaled so that the un-normalised 0..255 input
    (MADNet feeds raw pixel values, Nets/MadNet.py:56-66) produces KITTI-like disparities
    (mean ~15 px) instead of the hundreds of pixels a raw Xavier net gives."""
    w = xavier_weights(shapes, seed)
    for k in w:
        if k.endswith("conv1/weights") and "pyramid" in k:
            w[k] = (w[k] * conv1_gain).astype(np.float32)
        if k == "model/prediction/weights":        # DispNet: random nets predict ~0.3 px; scale to tens of px
            w[k] = (w[k] * dispnet_pred_gain).astype(np.float32)
    return w
