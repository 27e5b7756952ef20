"""CPU ORACLE (test infrastructure, NOT product code) -- MADNet graph + online step.

Restates Nets/MadNet.py (whole file) and the per-frame loop body of
Stereo_Online_Adaptation.py:178-253 on torch-CPU with autograd.  WIRING PINNED (round 4): tests/test_ref_graph.py holds this file's forward / loss / gradients to what the reference's own
graph code computes when oracle/ref_graph.py executes it under oracle/tf_shim; the TF library kernels' arithmetic stays
"parity unpinned" (oracle/tf_ops.py header).

Weights are a dict {TF variable name -> torch tensor} using the names the reference
graph creates under the driver's outer scope 'model/' (SURVEY App. C).
"""
import torch
from . import tf_ops as T

PYR_CH = [(3, 16, 2), (16, 16, 1), (16, 32, 2), (32, 32, 1), (32, 64, 2), (64, 64, 1),
          (64, 96, 2), (96, 96, 1), (96, 128, 2), (128, 128, 1), (128, 192, 2), (192, 192, 1)]
EST_CH = [128, 128, 96, 64, 32, 1]
CTX = [(128, 1), (128, 2), (128, 4), (96, 8), (64, 16), (32, 1), (1, 1)]
LEVEL_FEAT = {6: 12, 5: 10, 4: 8, 3: 6, 2: 4}      # level k -> conv index feeding it
LEVEL_C = {6: 192, 5: 128, 4: 96, 3: 64, 2: 32}


def variable_shapes(radius_d=2, stride=1):
    """Ordered {name: shape} of every trainable variable MADNet creates
    (Nets/MadNet.py:85-118,127-166,180-249; HWIO conv weights + biases)."""
    D = len(range(-radius_d, radius_d + 1, stride))
    out = {}
    for i, (ci, co, _) in enumerate(PYR_CH):
        out["model/gc-read-pyramid/conv%d/weights" % (i + 1)] = (3, 3, ci, co)
        out["model/gc-read-pyramid/conv%d/biases" % (i + 1)] = (co,)
    for k in (6, 5, 4, 3, 2):
        cin = LEVEL_C[k] + D + (0 if k == 6 else 1)
        for j, co in enumerate(EST_CH):
            out["model/G%d/fgc-volume-filtering-%d/disp-%d/weights" % (k, k, j + 1)] = (3, 3, cin, co)
            out["model/G%d/fgc-volume-filtering-%d/disp-%d/biases" % (k, k, j + 1)] = (co,)
            cin = co
    cin = LEVEL_C[2] + 1
    for j, (co, _) in enumerate(CTX):
        out["model/context-%d/weights" % (j + 1)] = (3, 3, cin, co)
        out["model/context-%d/biases" % (j + 1)] = (co,)
        cin = co
    return out


def layer_variables():
    """layer key -> [weights name, biases name] as StereoNet._add_to_layers records it
    (Nets/Stereo_net.py:54-79): the reused right tower's keys map to []."""
    m = {}
    for i in range(12):
        base = "model/gc-read-pyramid/conv%d/" % (i + 1)
        m["left/conv%d" % (i + 1)] = [base + "weights", base + "biases"]
        m["right/conv%d" % (i + 1)] = []
    for k in (6, 5, 4, 3, 2):
        for j in range(6):
            base = "model/G%d/fgc-volume-filtering-%d/disp-%d/" % (k, k, j + 1)
            m["fgc-volume-filtering-%d/disp%d" % (k, j + 1)] = [base + "weights", base + "biases"]
    for j in range(7):
        base = "model/context-%d/" % (j + 1)
        m["context%d" % (j + 1)] = [base + "weights", base + "biases"]
    return m


def forward(wts, left, right, bulkhead=False, radius_d=2, stride=1, warping=True, want_layers=False):
    """MadNet._preprocess_inputs + _build_network (Nets/MadNet.py:56-66,251-364).

    left/right: [B,H,W,3] float (0..255, no normalisation).  Returns the list of 6
    full-resolution disparities [d6,d5,d4,d3,d2ctx,final] (+ the layer dict)."""
    dt = left.dtype
    H0, W0 = left.shape[1], left.shape[2]
    L = T.pad_image(left, 64)
    R = T.pad_image(right, 64)
    Hp, Wp = L.shape[1], L.shape[2]
    layers = {}

    def pyramid(x, prefix):
        feats = []
        for i, (_, _, s) in enumerate(PYR_CH):
            w = wts["model/gc-read-pyramid/conv%d/weights" % (i + 1)]
            b = wts["model/gc-read-pyramid/conv%d/biases" % (i + 1)]
            x = T.conv2d(x, w, b, stride=s, alpha=0.2)
            layers["%s/conv%d" % (prefix, i + 1)] = x
            feats.append(x)
        return feats

    fl = pyramid(L, "left")
    fr = pyramid(R, "right")

    def make_disp(V):   # MadNet._make_disp (MadNet.py:68-71): relu BEFORE resize
        op = T.resize_bilinear(torch.relu(V * -20.0), Hp, Wp)
        return T.center_crop(op, H0, W0)

    def estimator(k, volume):
        x = volume
        for j in range(6):
            base = "model/G%d/fgc-volume-filtering-%d/disp-%d/" % (k, k, j + 1)
            x = T.conv2d(x, wts[base + "weights"], wts[base + "biases"], alpha=(0.2 if j < 5 else 1.0))
            layers["fgc-volume-filtering-%d/disp%d" % (k, j + 1)] = x
        return x

    disparities = []
    u = None
    V = None
    for k in (6, 5, 4, 3, 2):
        li = LEVEL_FEAT[k] - 1
        left_k, right_k = fl[li], fr[li]
        if k != 6 and warping:
            right_k = T.linear_warp(right_k, u)
        corr = T.correlation(left_k, right_k, radius_d, stride)
        dsi = torch.cat([left_k, corr], dim=-1)                       # MadNet.py:370-375
        vol = dsi if u is None else torch.cat([dsi, u], dim=-1)       # MadNet.py:77-80
        V = estimator(k, vol)
        if k != 2:
            disparities.append(make_disp(V))
            sc = 2 ** (k - 1)
            u = T.resize_bilinear(V, Hp // sc, Wp // sc) * 20.0 / sc  # MadNet.py:274
            if bulkhead:
                u = u.detach()                                        # MadNet.py:275-276
    V2_init = V
    x = torch.cat([fl[3], V2_init], dim=-1)                           # MadNet.py:123
    for j, (_, rate) in enumerate(CTX):
        base = "model/context-%d/" % (j + 1)
        x = T.conv2d(x, wts[base + "weights"], wts[base + "biases"], dilation=rate,
                     alpha=(0.2 if j < 6 else 1.0))
        layers["context%d" % (j + 1)] = x
    final_disp = V2_init + x                                          # MadNet.py:168
    layers["final_disp"] = final_disp
    disparities.append(make_disp(final_disp))
    resc = torch.relu(T.resize_bilinear(final_disp, Hp, Wp) * -20.0)  # MadNet.py:362: relu AFTER resize
    resc = T.center_crop(resc, H0, W0)
    layers["rescaled_prediction"] = resc
    disparities.append(resc)
    if want_layers:
        return disparities, layers
    return disparities


def momentum_update(wts, accum, grads, lr, momentum=0.9):
    """tf.train.MomentumOptimizer: accum = m*accum + g ; var -= lr*accum."""
    with torch.no_grad():
        for name, g in grads.items():
            accum[name].mul_(momentum).add_(g)
            wts[name].sub_(lr * accum[name])


def step(wts, accum, left, right, gt, mode="FULL", block_vars=None, block_index=None,
         lr=1e-4, radius_d=2, stride=1, loss="reprojection", proxy=None, reprojection_scale=1, warping=True, adam=None):
    """One iteration of the loop body Stereo_Online_Adaptation.py:178-253 (device part):
    ONE forward with pre-update weights, full-res loss + EPE/bad3, the selected
    backward and the momentum update.  mode in NONE/FULL/MAD.  For MAD, block_index is
    the sampled block (prediction index) and block_vars its variable-name list.
    adam = {"m": {...}, "v": {...}, "state": [beta1_power, beta2_power]}: the live demo's optimizer instead of momentum
    (Demo/demo_model.py:164 tf.train.AdamOptimizer(lr); one optimizer object serves all of the demo's train ops (:121-142),
    so the slots are per variable and the ONE pair of beta powers advances with every executed train op)."""
    names = list(wts.keys())
    for n in names:
        wts[n].requires_grad_(mode != "NONE")
    bulk = (mode == "MAD")
    disps = forward(wts, left, right, bulkhead=bulk, radius_d=radius_d, stride=stride, warping=warping)
    # loss="proxy": the continual-adaptation variant (Stereo_Continual_Adaptation.py:75,112): mean_l1 against proxy labels,
    # weight 0.01 on the full-resolution loss, 0.1 on a MAD block's loss
    full_loss = T.reprojection_loss(disps[-1], left, right) if loss == "reprojection" else T.proxy_loss(disps[-1], proxy, 0.01)
    epe, bad3 = T.validation_metrics(disps[-1].detach(), gt)
    grads = {}
    if mode == "FULL":
        gl = torch.autograd.grad(full_loss, [wts[n] for n in names], allow_unused=True)
        grads = {n: g for n, g in zip(names, gl) if g is not None}
    elif mode == "MAD":
        p = disps[block_index]
        mult = float(left.shape[1] // p.shape[1])                 # Stereo_Online_Adaptation.py:102-103
        # inputs_modules = scale_tensor(., reprojectionScale) (Stereo_Online_Adaptation.py:22-23,91-95); `mult` stays the ratio to
        # the UNSCALED frame (:102), i.e. 1 for the full-resolution predictions
        s = int(reprojection_scale)
        ls = T.resize_bilinear(left, left.shape[1] // s, left.shape[2] // s) if s != 1 else left
        rs = T.resize_bilinear(right, right.shape[1] // s, right.shape[2] // s) if s != 1 else right
        p = T.resize_bilinear(p, ls.shape[1], ls.shape[2]) * mult
        loss_k = T.reprojection_loss(p, ls, rs) if loss == "reprojection" else T.proxy_loss(p, proxy, 0.1)
        vs = [n for n in block_vars]
        gl = torch.autograd.grad(loss_k, [wts[n] for n in vs], allow_unused=True)
        grads = {n: g for n, g in zip(vs, gl) if g is not None}
    for n in names:
        wts[n].requires_grad_(False)
    out = {"loss": float(full_loss.detach()), "epe": float(epe), "bad3": float(bad3),
           "disparity": disps[-1].detach(), "grads": {k: v.detach() for k, v in grads.items()}}
    if grads and adam is not None:
        with torch.no_grad():
            for n, g in grads.items():
                T.adam_update(wts[n], adam["m"][n], adam["v"][n], g, adam["state"], lr)
            st = adam["state"]
            st[0] = float(torch.tensor(st[0], dtype=torch.float32) * torch.tensor(0.9, dtype=torch.float32))
            st[1] = float(torch.tensor(st[1], dtype=torch.float32) * torch.tensor(0.999, dtype=torch.float32))
    elif grads:
        momentum_update(wts, accum, grads, lr)
    return out



def train_step(wts, adam_m, adam_v, adam_state, left, right, gt, lr=1e-4, loss_weights=None, max_disp=192.0,
               radius_d=2, stride=1):
    """One iteration of the offline training loop Train.py:94-102,125-140 (device part): forward without bulkhead, the
    multi-scale supervised mean_l1 (weights from full to lowest resolution = disparities[-1], [-2], ...), gradients of every
    variable, Adam(lr, 0.9).  adam_state = [beta1_power, beta2_power] (python floats, updated in place)."""
    names = list(wts.keys())
    for n in names:
        wts[n].requires_grad_(True)
    disps = forward(wts, left, right, bulkhead=False, radius_d=radius_d, stride=stride)
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
        b1p = torch.tensor(adam_state[0], dtype=torch.float32) * torch.tensor(0.9, dtype=torch.float32)
        b2p = torch.tensor(adam_state[1], dtype=torch.float32) * torch.tensor(0.999, dtype=torch.float32)
        adam_state[0], adam_state[1] = float(b1p), float(b2p)
    return {"loss": float(total.detach()), "losses": [float(x.detach()) for x in parts], "disparity": disps[-1].detach(), "grads": grads}
