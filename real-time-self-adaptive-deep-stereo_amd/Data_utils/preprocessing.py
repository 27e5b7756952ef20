"""Hot image-space functions of the reference (Data_utils/preprocessing.py:7-29,121-230,269-277) on
torch GPU tensors, backed by the HIP kernels: pad_image, warp_image / bilinear_sampler (horizontal
disparity form), rescale_image, resize_to_prediction.  Training-time augmentation / colour mapping
(random_crop, augment, colorize_img) are outside the hot path and not provided."""
import torch

from madnet_hip import _ffi, ops


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


def pad_image(immy, down_factor=256, dynamic=False):
    """REFLECT-pad H,W up to a multiple of down_factor (before=(new-old)//2, after=(new-old+1)//2)."""
    B, H, W, Cc = immy.shape
    nh = H if H % down_factor == 0 else (H // down_factor + 1) * down_factor
    nw = W if W % down_factor == 0 else (W // down_factor + 1) * down_factor
    out = torch.empty(B, nh, nw, Cc, device=immy.device)
    ops.pad_reflect(_ffi.lib(), immy.contiguous().float(), out, (nh - H) // 2, (nw - W) // 2, stream=_stream(immy))
    return out


class _ResizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, oh, ow):
        B, H, W, Cc = x.shape
        assert Cc == 1, "HIP resize handles single-channel maps (disparities)"
        xin = x.contiguous().view(B, H, W)
        out = torch.empty(B, oh, ow, device=x.device)
        ops.resize_fwd(_ffi.lib(), xin, out, oh, ow, stream=_stream(x))
        ctx.save_for_backward(xin)
        ctx.size = (oh, ow)
        return out[..., None]

    @staticmethod
    def backward(ctx, g):
        (xin,) = ctx.saved_tensors
        oh, ow = ctx.size
        dx = torch.empty_like(xin)
        ops.resize_bwd(_ffi.lib(), g.contiguous().view(g.shape[0], oh, ow), xin, dx, oh, ow, stream=_stream(xin))
        return dx[..., None], None, None


def rescale_image(img, out_shape):
    """tf.image.resize_images(bilinear), TF1 legacy kernel (no half-pixel centres)."""
    oh, ow = int(out_shape[0]), int(out_shape[1])
    if (img.shape[1], img.shape[2]) == (oh, ow):
        return img
    return _ResizeFn.apply(img, oh, ow)


def resize_to_prediction(x, pred):
    return rescale_image(x, pred.shape[1:3])


def warp_image(img, flow):
    """Right image warped to the left view by the disparity `flow` [B,H,W,1]: coords (x - d, y),
    4-tap bilinear sampling with indices clamped to the border and UN-masked weights (the code's
    behaviour, not its docstring: SURVEY App. D.7).  img: [B,H,W,3]."""
    from Losses import loss_factory
    return loss_factory._warp_only(img, flow)


def bilinear_sampler(imgs, coords):
    """General form is not on the hot path; only the disparity form (coords = (x - d, y)) is built."""
    raise NotImplementedError("use warp_image(img, disparity); arbitrary coordinate sampling is not on the hot path")
