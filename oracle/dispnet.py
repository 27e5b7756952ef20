"""CPU ORACLE (test infrastructure, NOT product code) -- DispNet-C graph + FULL adaptation step.

Restates Nets/DispNet.py (whole file) on torch-CPU with the TF 1.12 op semantics of oracle/tf_ops.py.
WIRING PINNED (round 4): tests/test_ref_graph.py holds forward / loss / gradients to the reference's own graph code executed under
oracle/tf_shim (oracle/ref_graph.py); the TF library kernels' arithmetic stays "parity unpinned" (oracle/tf_ops.py header).  Variable names follow the reference
scopes under the driver's 'model/' (SURVEY App. C): default bias name is 'bias' (sharedLayers.py:54,80).
"""
import torch
from . import tf_ops as T

MAX_DISP = 40
ALPHA = 0.1          # default activation of sharedLayers.conv2d / conv2d_transpose


def variable_shapes():
    """Ordered {name: shape}; conv weights HWIO, transposed-conv weights [kh,kw,Cout,Cin] (sharedLayers.py:80-92)."""
    out = {}

    def conv(name, k, ci, co):
        out["model/%s/weights" % name] = (k, k, ci, co)
        out["model/%s/bias" % name] = (co,)

    def deconv(name, co, ci):
        out["model/%s/weights" % name] = (4, 4, co, ci)
        out["model/%s/bias" % name] = (co,)

    conv("conv1", 7, 3, 64); conv("conv2", 5, 64, 128); conv("conv_redir", 1, 128, 64)
    conv("conv3", 5, 2 * MAX_DISP + 1 + 64, 256); conv("conv3/1", 3, 256, 256)
    conv("conv4", 3, 256, 512); conv("conv4/1", 3, 512, 512)
    conv("conv5", 3, 512, 512); conv("conv5/1", 3, 512, 512)
    conv("conv6", 3, 512, 1024); conv("conv6/1", 3, 1024, 1024)
    for name, cin, cout, skip in (("up5", 1024, 512, 512), ("up4", 512, 256, 512), ("up3", 256, 128, 256),
                                  ("up2", 128, 64, 128), ("up1", 64, 32, 64)):
        deconv(name + "/deconv", cout, cin)
        conv(name + "/predict", 3, cin, 1)
        deconv(name + "/up_predict", 1, 1)
        conv(name + "/concat", 3, cout + skip + 1, cout)
    conv("prediction", 3, 32, 1)
    return out


def forward(wts, left, right, want_layers=False):
    """DispNet._preprocess_inputs + _build_network (correlation=True), DispNet.py:59-152.
    Returns the 7 disparities [up5..up1 predicts, prediction, rescaled_prediction]."""
    H0, W0 = left.shape[1], left.shape[2]
    dt = left.dtype
    sub = torch.tensor(100.0 / 255, dtype=dt)
    L = T.pad_image(left / 255.0 - sub, 64)
    R = T.pad_image(right / 255.0 - sub, 64)
    Hp, Wp = L.shape[1], L.shape[2]
    layers, disps = {}, []

    def W(n):
        return wts["model/%s/weights" % n], wts["model/%s/bias" % n]

    def conv(x, n, stride=1, alpha=ALPHA):
        w, b = W(n)
        return T.conv2d(x, w, b, stride=stride, alpha=alpha)

    def make_disp(op):     # DispNet._make_disp (DispNet.py:39-43)
        scale = float(Wp) / float(op.shape[2])
        o = T.resize_bilinear(torch.relu(op * scale), Hp, Wp)
        return T.center_crop(o, H0, W0)

    c1a, c1b = conv(L, "conv1", 2), conv(R, "conv1", 2)
    c2a, c2b = conv(c1a, "conv2", 2), conv(c1b, "conv2", 2)
    redir = conv(c2a, "conv_redir")
    corr = T.correlation(c2a, c2b, MAX_DISP)
    c3 = conv(torch.cat([corr, redir], -1), "conv3", 2)
    c31 = conv(c3, "conv3/1")
    c4 = conv(c31, "conv4", 2); c41 = conv(c4, "conv4/1")
    c5 = conv(c41, "conv5", 2); c51 = conv(c5, "conv5/1")
    c6 = conv(c51, "conv6", 2); c61 = conv(c6, "conv6/1")
    layers.update({"conv1a": c1a, "conv1b": c1b, "conv2a": c2a, "conv2b": c2b, "conv_redir": redir, "corr": corr,
                   "conv3": c3, "conv3/1": c31, "conv4": c4, "conv4/1": c41, "conv5": c5, "conv5/1": c51,
                   "conv6": c6, "conv6/1": c61})

    def up(bottom, skip, name):     # DispNet._upsampling_block (DispNet.py:45-57)
        w, b = W(name + "/deconv")
        dec = T.conv2d_transpose(bottom, w, b, stride=2, alpha=ALPHA)
        pred = conv(bottom, name + "/predict", alpha=1.0)
        disps.append(make_disp(pred))
        w, b = W(name + "/up_predict")
        upp = T.conv2d_transpose(pred, w, b, stride=2, alpha=1.0)
        out = conv(torch.cat([skip, dec, upp], -1), name + "/concat", alpha=1.0)
        layers.update({name + "/deconv": dec, name + "/predict": pred, name + "/up_predict": upp, name + "/concat": out})
        return out

    u5 = up(c61, c51, "up5"); u4 = up(u5, c41, "up4"); u3 = up(u4, c31, "up3")
    u2 = up(u3, c2a, "up2"); u1 = up(u2, c1a, "up1")
    pred = conv(u1, "prediction", alpha=1.0)
    layers["prediction"] = pred
    disps.append(make_disp(pred))
    resc = T.center_crop(T.resize_bilinear(pred, Hp, Wp) * 2.0, H0, W0)      # DispNet.py:149-151 (no relu)
    layers["rescaled_prediction"] = resc
    disps.append(resc)
    return (disps, layers) if want_layers else disps


def step(wts, accum, left, right, gt, mode="FULL", lr=1e-4):
    """Loop body of Stereo_Online_Adaptation.py:178-253 for DispNet (modes NONE / FULL; the shipped
    dispnet_full.json has 5 groups for 6 predictions so the MAD assert :97 fails in the reference)."""
    from .madnet import momentum_update
    names = list(wts.keys())
    for n in names:
        wts[n].requires_grad_(mode == "FULL")
    disps = forward(wts, left, right)
    loss = T.reprojection_loss(disps[-1], left, right)
    epe, bad3 = T.validation_metrics(disps[-1].detach(), gt)
    grads = {}
    if mode == "FULL":
        gl = torch.autograd.grad(loss, [wts[n] for n in names], allow_unused=True)
        grads = {n: g for n, g in zip(names, gl) if g is not None}
    for n in names:
        wts[n].requires_grad_(False)
    out = {"loss": float(loss.detach()), "epe": float(epe), "bad3": float(bad3), "disparity": disps[-1].detach(),
           "grads": {k: v.detach() for k, v in grads.items()}}
    if grads:
        momentum_update(wts, accum, grads, lr)
    return out


def train_step(wts, adam_m, adam_v, adam_state, left, right, gt, lr=1e-4, loss_weights=None, max_disp=192.0):
    """One iteration of Train.py:94-102,125-140 for DispNet (its default model): multi-scale supervised mean_l1 over the 7
    predictions (weights from disparities[-1] = rescaled_prediction down to up5/predict), every variable, Adam(lr, 0.9)."""
    names = list(wts.keys())
    for n in names:
        wts[n].requires_grad_(True)
    disps = forward(wts, left, right)
    lw = list(loss_weights) if loss_weights is not None else [1.0] * 10
    parts = [T.supervised_loss(disps[-(i + 1)], gt, lw[i], max_disp) for i in range(len(disps))]
    total = sum(parts)
    gl = torch.autograd.grad(total, [wts[n] for n in names], allow_unused=True)
    for n in names:
        wts[n].requires_grad_(False)
    grads = {n: g.detach() for n, g in zip(names, gl) if g is not None}
    with torch.no_grad():
        for n, g in grads.items():
            T.adam_update(wts[n], adam_m[n], adam_v[n], g, adam_state, lr)
        adam_state[0] = float(torch.tensor(adam_state[0], dtype=torch.float32) * torch.tensor(0.9, dtype=torch.float32))
        adam_state[1] = float(torch.tensor(adam_state[1], dtype=torch.float32) * torch.tensor(0.999, dtype=torch.float32))
    return {"loss": float(total.detach()), "losses": [float(x.detach()) for x in parts], "disparity": disps[-1].detach(), "grads": grads}

