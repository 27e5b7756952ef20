"""Reprojection loss of the online-adaptation path with the reference's factory API
(Losses/loss_factory.py:353-395): get_reprojection_loss('mean_SSIM_l1', ...)(disparities, inputs).
Only the loss the online script requests (Stereo_Online_Adaptation.py:70,107) is on the hot path;
the supervised / proxy / other photometric variants are out of scope."""
import torch

from madnet_hip import _ffi, ops


def _lib():
    return _ffi.lib()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


class _ReprojFn(torch.autograd.Function):
    """loss = 0.85*mean(SSIM3x3(warp(right/256, d), left/256)) + 0.15*mean|.|; gradient w.r.t. d."""

    @staticmethod
    def forward(ctx, disp, left, right):
        lib = _lib()
        B, H, W, _ = disp.shape
        d = disp.contiguous().view(B, H, W)
        ws = torch.empty(lib.loss_ws_floats(B, H, W), device=disp.device)
        res = torch.zeros(4, device=disp.device)
        dd = torch.empty(B, H, W, device=disp.device)
        ops.reprojection_loss(lib, left.contiguous().float(), right.contiguous().float(), d, ws, res, dd, 1.0, stream=_stream(disp))
        ctx.save_for_backward(dd)
        return res[0]

    @staticmethod
    def backward(ctx, g):
        (dd,) = ctx.saved_tensors
        return (dd * g)[..., None], None, None


ALL_LOSSES = {'mean_SSIM_l1': None}


def get_reprojection_loss(reconstruction_loss, multiScale=False, logs=False, weights=None, reduced=True):
    if reconstruction_loss not in ALL_LOSSES.keys():
        print('Unrecognized loss function, pick one among: {}'.format(ALL_LOSSES.keys()))
        raise Exception('Unknown loss function selected')
    if weights is None:
        weights = [1] * 10

    def compute_loss(disparities, inputs):
        from Data_utils import preprocessing
        left, right = inputs['left'], inputs['right']
        accumulator = []
        disp_to_test = len(disparities) if multiScale else 1
        for i in range(disp_to_test):
            current_disp = disparities[-(i + 1)]
            scale = float(left.shape[2]) / float(current_disp.shape[2])
            resized = preprocessing.resize_to_prediction(current_disp, left) * scale
            accumulator.append(weights[i] * _ReprojFn.apply(resized, left, right))
        return sum(accumulator) if reduced else accumulator
    return compute_loss
