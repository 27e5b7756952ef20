"""TEST INFRASTRUCTURE (CPU oracle): runs the REFERENCE's own loss builders get_supervised_loss / get_proxy_loss
(/root/reference/Losses/loss_factory.py:256-351, with mean_l1 :28-38 and preprocessing.resize_to_prediction) eagerly under oracle/tf_shim's stand-in
`tensorflow`, on seeded inputs, and dumps values + gradients w.r.t. every prediction.  This pins oracle/tf_ops.py::supervised_loss / proxy_loss -- the
loss heads of the offline training step (Train.py:100) and of the continual-adaptation variant (Stereo_Continual_Adaptation.py:75,112) -- to the
reference source the way oracle/ref_graph.py pins the networks (VERDICT r04 missing 4).  Subprocess with a path of its own, nothing under
/root/reference is written.

    python oracle/ref_losses.py --inputs in.npz --out out.npz

in.npz : left, right, target, proxy [B,H,W,C]; pred_0 .. pred_{n-1} = the network's disparities list (coarse first, rescaled_prediction last), each
         [B,h_i,W_i,1]; sup_weights [n] (Train.py --lossWeights), max_disp.
out.npz: sup_loss (multiScale=True, reduced) ; sup_parts [n] (reduced=False) ; sup_grad_<i> = d sup_loss / d pred_i ;
         sup1_loss / sup1_grad (multiScale=False: the last prediction only) ;
         proxy_loss / proxy_grad (get_proxy_loss('mean_l1') defaults: weights 0.01, last prediction) ;
         proxy01_loss / proxy01_grad_<i> (weights 0.1: the MAD blocks' losses, Stereo_Continual_Adaptation.py:112, on prediction i alone)."""
import argparse
import os
import sys

REF = os.environ.get("MADNET_REFERENCE_ROOT", "/root/reference")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inputs", required=True)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    sys.dont_write_bytecode = True
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:] = [os.path.join(here, "tf_shim"), REF] + [p for p in sys.path if p and os.path.abspath(p) != os.path.dirname(here)
                                                            and "real-time-self-adaptive-deep-stereo_amd" not in p] + [os.path.dirname(here)]
    import numpy as np
    import torch
    import tensorflow as tf
    assert tf.__version__.endswith("shim") and tf.__file__.startswith(here)
    from Losses import loss_factory
    assert os.path.abspath(loss_factory.__file__).startswith(os.path.abspath(REF)), loss_factory.__file__

    z = np.load(a.inputs)
    n = sum(1 for k in z.files if k.startswith("pred_"))
    inputs = {k: tf.constant(z[k], tf.float32) for k in ("left", "right", "target", "proxy")}
    weights = [float(w) for w in z["sup_weights"]]
    max_disp = float(z["max_disp"])

    def preds():
        out = []
        for i in range(n):
            t = tf.constant(z["pred_%d" % i], tf.float32)
            t.t.requires_grad_(True)
            out.append(t)
        return out

    out = {}
    p = preds()
    loss = loss_factory.get_supervised_loss("mean_l1", multiScale=True, logs=False, weights=weights, max_disp=max_disp)(p, inputs)      # Train.py:100
    out["sup_loss"] = np.float32(loss.t.detach().numpy())
    gs = torch.autograd.grad(loss.t, [q.t for q in p], allow_unused=True)
    for i, g in enumerate(gs):
        out["sup_grad_%d" % i] = g.numpy() if g is not None else np.zeros_like(z["pred_%d" % i])
    p = preds()
    parts = loss_factory.get_supervised_loss("mean_l1", multiScale=True, logs=False, weights=weights, reduced=False, max_disp=max_disp)(p, inputs)
    out["sup_parts"] = np.array([float(q.t.detach().numpy()) for q in parts], dtype=np.float32)      # parts[i] belongs to disparities[-(i+1)]
    p = preds()
    loss = loss_factory.get_supervised_loss("mean_l1", multiScale=False, max_disp=max_disp)(p, inputs)
    out["sup1_loss"] = np.float32(loss.t.detach().numpy())
    out["sup1_grad"] = torch.autograd.grad(loss.t, [p[-1].t])[0].numpy()
    p = preds()
    loss = loss_factory.get_proxy_loss("mean_l1")(p, inputs)                                         # Stereo_Continual_Adaptation.py:75
    out["proxy_loss"] = np.float32(loss.t.detach().numpy())
    out["proxy_grad"] = torch.autograd.grad(loss.t, [p[-1].t])[0].numpy()
    for i in range(n):
        p = preds()
        loss = loss_factory.get_proxy_loss("mean_l1", weights=[0.1] * 10, reduced=True)([p[i]], inputs)     # :112, one block's prediction
        out["proxy01_loss_%d" % i] = np.float32(loss.t.detach().numpy())
        out["proxy01_grad_%d" % i] = torch.autograd.grad(loss.t, [p[i].t])[0].numpy()
    np.savez(a.out, **out)


if __name__ == "__main__":
    main()
