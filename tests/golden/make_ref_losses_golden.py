"""Mints tests/golden/ref_losses.npz by EXECUTING the reference's own get_supervised_loss / get_proxy_loss (/root/reference/Losses/loss_factory.py:256-351)
under oracle/tf_shim's eager stand-in for tensorflow (oracle/ref_losses.py, a subprocess), on seeded inputs -- in the build container, where /root/reference
exists.  tests/test_ref_graph.py checks oracle/tf_ops.py::supervised_loss / proxy_loss against it everywhere and mh_supervised_loss / mh_proxy_loss on the MI355X.

    python tests/golden/make_ref_losses_golden.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = os.environ.get("MADNET_REFERENCE_ROOT", "/root/reference")
H, W, NPRED = 36, 52, 6
MAX_DISP = 192.0
SUP_WEIGHTS = [1.0, 0.8, 0.6, 0.4, 0.2, 0.1]          # Train.py --lossWeights: "from full to lower res" (weights[i] goes with disparities[-(i+1)])


def inputs():
    """seeded: six full-resolution predictions (what _make_disp returns), a target with invalid (0) and over-range (>= max_disp) pixels, proxies with
    invalid (<= 0) and over-range (>= 192) pixels"""
    rs = np.random.RandomState(7)
    z = {"left": rs.uniform(0, 255, (1, H, W, 3)).astype(np.float32), "right": rs.uniform(0, 255, (1, H, W, 3)).astype(np.float32)}
    t = rs.uniform(1, 150, (1, H, W, 1)).astype(np.float32)
    t[rs.uniform(size=t.shape) < 0.3] = 0.0
    t[rs.uniform(size=t.shape) < 0.05] = 200.0
    z["target"] = t
    q = rs.uniform(1, 150, (1, H, W, 1)).astype(np.float32)
    q[rs.uniform(size=q.shape) < 0.2] = 0.0
    q[rs.uniform(size=q.shape) < 0.05] = -3.0
    q[rs.uniform(size=q.shape) < 0.05] = 192.0
    z["proxy"] = q
    for i in range(NPRED):
        z["pred_%d" % i] = rs.uniform(0, 160, (1, H, W, 1)).astype(np.float32)
    z["sup_weights"] = np.array(SUP_WEIGHTS, dtype=np.float32)
    z["max_disp"] = np.float32(MAX_DISP)
    return z


def run_reference():
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.npz"), os.path.join(td, "out.npz")
        np.savez(fin, **inputs())
        rc = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_losses.py"), "--inputs", fin, "--out", fout], capture_output=True, text=True,
                            env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
        if rc.returncode != 0:
            raise RuntimeError("oracle/ref_losses.py failed:\n%s" % rc.stderr[-3000:])
        z = np.load(fout)
        return {k: z[k] for k in z.files}


if __name__ == "__main__":
    ref = run_reference()
    path = os.path.join(GOLD, "ref_losses.npz")
    np.savez_compressed(path, **ref)
    print("%d arrays, %.1f KB; sup_loss %.6f (parts %s), proxy_loss %.6f" % (len(ref), os.path.getsize(path) / 1024.0, float(ref["sup_loss"]), ref["sup_parts"], float(ref["proxy_loss"])))
