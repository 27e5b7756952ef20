"""Hot image-space functions of the reference (Data_utils/preprocessing.py:7-29,121-230,269-277) on torch tensors (storage
only), backed by the HIP kernels: pad_image, bilinear_sampler (general form) / warp_image, rescale_image,
resize_to_prediction -- same names, argument meaning and autograd behaviour (gradients flow to the sampled coordinates / the
disparity and to the images; tf.floor contributes none).  Training-time augmentation lives in Data_utils/data_reader.py;
colour mapping (colorize_img) is outside the hot path and not provided."""
import torch

from madnet_hip import _ffi, ops


def _lib():
    return _ffi.lib()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


def pad_image(immy, down_factor=256, dynamic=False):
    """REFLECT-pad H,W up to a multiple of down_factor (before=(new-old)//2, after=(new-old+1)//2)  (preprocessing.py:7-29)."""
    B, H, W, Cc = immy.shape
    nh = H if H % down_factor == 0 else (H // down_factor + 1) * down_factor
    nw = W if W % down_factor == 0 else (W // down_factor + 1) * down_factor
    out = torch.empty(B, nh, nw, Cc, device=immy.device)
    ops.pad_reflect(_lib(), immy.contiguous().float(), out, (nh - H) // 2, (nw - W) // 2, stream=_stream(immy))
    return out


class _ResizeFn(torch.autograd.Function):
    """tf.image.resize_images(bilinear), TF1 legacy kernel, any channel count; gradient = ResizeBilinearGrad."""

    @staticmethod
    def forward(ctx, x, oh, ow):
        B, H, W, Cc = x.shape
        xin = x.contiguous().float()
        out = torch.empty(B, oh, ow, Cc, device=x.device)
        _lib().resize_image_fwd(ops._p(xin), ops._p(out), B, H, W, Cc, oh, ow, ops._p(_stream(x)))
        ctx.shape = (B, H, W, Cc, oh, ow)
        return out

    @staticmethod
    def backward(ctx, g):
        B, H, W, Cc, oh, ow = ctx.shape
        g = g.contiguous()
        dx = torch.empty(B, H, W, Cc, device=g.device)
        _lib().resize_image_bwd(ops._p(g), ops._p(dx), B, H, W, Cc, oh, ow, ops._p(_stream(g)))
        return dx, None, None


def rescale_image(img, out_shape):
    """preprocessing.rescale_image (preprocessing.py:269-273, FULLY_DIFFERENTIABLE = False): tf.image.resize_images(bilinear),
    the TF1 legacy kernel (no half-pixel centres).  Identity at equal size."""
    oh, ow = int(out_shape[0]), int(out_shape[1])
    if (img.shape[1], img.shape[2]) == (oh, ow):
        return img
    return _ResizeFn.apply(img, oh, ow)


def resize_to_prediction(x, pred):
    return rescale_image(x, pred.shape[1:3])


class _SamplerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, imgs, coords):
        imgs = imgs.contiguous().float(); coords = coords.contiguous().float()
        B, Hs, Ws, Cc = imgs.shape
        _, Ht, Wt, _ = coords.shape
        out = torch.empty(B, Ht, Wt, Cc, device=imgs.device)
        _lib().bilinear_sampler_fwd(ops._p(imgs), ops._p(coords), ops._p(out), B, Hs, Ws, Cc, Ht, Wt, ops._p(_stream(imgs)))
        ctx.save_for_backward(imgs, coords)
        return out

    @staticmethod
    def backward(ctx, g):
        imgs, coords = ctx.saved_tensors
        B, Hs, Ws, Cc = imgs.shape
        _, Ht, Wt, _ = coords.shape
        g = g.contiguous()
        dcoords = torch.empty_like(coords) if ctx.needs_input_grad[1] else None
        dimgs = torch.zeros_like(imgs) if ctx.needs_input_grad[0] else None
        if dcoords is None and dimgs is None:
            return None, None
        _lib().bilinear_sampler_bwd(ops._p(g), ops._p(imgs), ops._p(coords), ops._p(dcoords), ops._p(dimgs), B, Hs, Ws, Cc, Ht, Wt,
                                    ops._p(_stream(g)))
        return dimgs, dcoords


def bilinear_sampler(imgs, coords):
    """preprocessing.bilinear_sampler (preprocessing.py:121-199): imgs [B,Hs,Ws,C], coords [B,Ht,Wt,2] (x, y) -> [B,Ht,Wt,C].
    Indices are clamped to the border and the weights are NOT masked -- the code's behaviour, not its docstring ("points
    outside ... have value 0"): SURVEY App. D.7."""
    return _SamplerFn.apply(imgs, coords)


def warp_image(img, flow):
    """preprocessing.warp_image (preprocessing.py:201-230): coords = (x - flow, y), then bilinear_sampler.  img [B,H,W,C],
    flow [B,H,W,1] (for stereo: img = right image, flow = disparity aligned with the left one)."""
    B, H, W, _ = flow.shape
    xs = torch.arange(W, dtype=torch.float32, device=flow.device).view(1, 1, W, 1).expand(B, H, W, 1)
    ys = torch.arange(H, dtype=torch.float32, device=flow.device).view(1, H, 1, 1).expand(B, H, W, 1)
    coords = torch.cat([xs - flow, ys], dim=-1)
    return bilinear_sampler(img, coords)
