"""The three loss builders of the reference's factory (Losses/loss_factory.py) over the HIP loss kernels, same names / arguments / return shapes:

  get_reprojection_loss('mean_SSIM_l1', ...)(disparities, inputs)   :353-395  online adaptation (Stereo_Online_Adaptation.py:70,107) -> mh_reprojection_loss
  get_supervised_loss('mean_l1', ...)(disparities, inputs)          :256-302  offline training (Train.py:100)                        -> mh_supervised_loss
  get_proxy_loss('mean_l1', ...)(disparities, inputs)               :304-351  continual adaptation (Stereo_Continual_Adaptation.py:75,112) -> mh_proxy_loss

Each returns compute_loss(disparities, inputs) like the reference; the per-scale terms are torch.autograd.Functions whose backward is the gradient the kernel
wrote in the same launch.  Only the base losses the three scripts request by default are kernels ('mean_SSIM_l1' / 'mean_l1'); every other name of the
reference's SUPERVISED_LOSS / PIXELWISE_LOSSES tables raises (out of scope)."""
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


class _MaskedL1Fn(torch.autograd.Function):
    """weight * sum(valid * |pred - label|) / sum(valid) (mean_l1, loss_factory.py:28-38) with the validity rule of the builder:
    kind 'supervised': valid = !(label == 0 | label >= max_disp) (:285);  kind 'proxy': valid = !(label <= 0 | label >= 192) (:337).
    (The reference's mean_l1 divides by sum(valid) un-guarded; so do the kernels: an all-invalid label map gives the same NaN / Inf.)"""

    @staticmethod
    def forward(ctx, pred, label, kind, weight, max_disp):
        lib = _lib()
        B, H, W, _ = pred.shape
        p = pred.contiguous().float().view(B, H, W)
        t = label.contiguous().float().view(B, H, W)
        ws = torch.empty(int(lib.proxy_ws_floats(B, H, W)), device=pred.device)
        res = torch.zeros(4, device=pred.device)
        dp = torch.empty(B, H, W, device=pred.device)
        if kind == 'proxy':
            ops.proxy_loss(lib, p, t, ws, res, dpred=dp, weight=float(weight), stream=_stream(pred))
        else:
            ops.supervised_loss(lib, p, t, ws, res, dpred=dp, weight=float(weight), max_disp=float(max_disp), stream=_stream(pred))
        ctx.save_for_backward(dp)
        return res[0]

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        return (dp * g)[..., None], None, None, None, None


ALL_LOSSES = {'mean_SSIM_l1': None}
LABEL_LOSSES = {'mean_l1': None}           # the base loss Train.py / Stereo_Continual_Adaptation.py request (their --lossType default / hard-coded name)


def _label_loss(kind, name, multiScale, weights, reduced, max_disp):
    if name not in LABEL_LOSSES.keys():
        print('Unrecognized loss function, pick one among: {}'.format(LABEL_LOSSES.keys()))
        raise Exception('Unknown loss function selected')

    def compute_loss(disparities, inputs):
        from Data_utils import preprocessing
        left, targets = inputs['left'], inputs['target']
        labels = inputs['proxy'] if kind == 'proxy' else targets
        accumulator = []
        disp_to_test = len(disparities) if multiScale else 1
        for i in range(disp_to_test):
            current_disp = disparities[-(i + 1)]
            scale = float(left.shape[2]) / float(current_disp.shape[2])
            resized = preprocessing.resize_to_prediction(current_disp, targets) * scale
            accumulator.append(_MaskedL1Fn.apply(resized, labels, kind, weights[i], max_disp))
        return sum(accumulator) if reduced else accumulator
    return compute_loss


def get_supervised_loss(name, multiScale=False, logs=False, weights=None, reduced=True, max_disp=None):
    """loss_factory.py:256-302: weights default [1]*10 (weights[i] goes with disparities[-(i+1)]), max_disp default 1000; inputs needs 'left', 'target'."""
    return _label_loss('supervised', name, multiScale, [1] * 10 if weights is None else weights, reduced, 1000 if max_disp is None else max_disp)


def get_proxy_loss(name, multiScale=False, logs=False, weights=None, reduced=True, max_disp=None):
    """loss_factory.py:304-351: weights default [0.01]*10; the validity range is the reference's hard-coded (0, 192) whatever max_disp says (:337);
    inputs needs 'left', 'target' (its shape sizes the prediction), 'proxy'."""
    return _label_loss('proxy', name, multiScale, [0.01] * 10 if weights is None else weights, reduced, 192)


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
