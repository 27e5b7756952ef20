"""Which gradients of the product path's FULL step depend on the arrival order of float atomics?  The 'grad' part of the step (forward + loss + backward, no
update) is replayed N times from identical state on the MI355X and the flat gradient buffer is compared variable by variable, bit for bit; the feature-gradient
buffers (the warp gradient's scatter target) and the step's disparity likewise.   python scripts/exp/det_probe.py [--runs 6] [--precision mixed] [--model madnet]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=6)
    ap.add_argument("--precision", default="mixed")
    ap.add_argument("--model", default="madnet")
    ap.add_argument("--height", type=int, default=375)
    ap.add_argument("--width", type=int, default=1242)
    a = ap.parse_args()
    import torch
    from madnet_hip import _ffi, engine as E, dispnet_engine as DE, synthetic as S
    lib = _ffi.lib()
    dn = a.model == "dispnet"
    wn = S.calibrated_weights(dict(DE.dispnet_manifest() if dn else E.madnet_manifest()), 1)
    l, r, gt = S.make_pair(a.height, a.width)
    eng = (DE.DispNetEngine if dn else E.MadNetEngine)(lib, a.height, a.width, B=1, device="cuda:0", weights=wn, precision=a.precision)
    eng.set_inputs(l, r, gt[..., 0])
    plan = eng.build_plan("FULL", lr=1e-4, part="grad")
    st = torch.cuda.Stream()
    snaps = []
    with torch.cuda.stream(st):
        plan.run(lib, st.cuda_stream)
        st.synchronize()
        plan.capture(lib, st.cuda_stream)
        for _ in range(a.runs):
            plan.launch(lib, st.cuda_stream)
            st.synchronize()
            snaps.append((eng.params.g.clone(), eng.pred.clone(), eng.dF_levels.clone() if hasattr(eng, "dF_levels") else None))
    g0 = snaps[0][0]
    bad = {}
    for name, _ in eng.params.manifest:
        off, n = eng.params.offset[name], eng.params.numel(name)
        diffs = [int((sn[0][off:off + n] != g0[off:off + n]).sum().item()) for sn in snaps[1:]]
        if any(diffs):
            ref = g0[off:off + n].abs().max().item()
            worst = max((sn[0][off:off + n] - g0[off:off + n]).abs().max().item() for sn in snaps[1:])
            bad[name] = (n, max(diffs), worst / max(ref, 1e-30))
    print("%s %s %dx%d: %d replays of the 'grad' plan" % (a.model, a.precision, a.width, a.height, a.runs))
    print("disparity bit-identical: %s" % all(torch.equal(sn[1], snaps[0][1]) for sn in snaps[1:]))
    if snaps[0][2] is not None:
        print("feature-gradient buffers bit-identical: %s" % all(torch.equal(sn[2], snaps[0][2]) for sn in snaps[1:]))
    print("whole gradient buffer bit-identical: %s" % all(torch.equal(sn[0], g0) for sn in snaps[1:]))
    for k in sorted(bad):
        print("  NOT reproducible: %-60s %8d elements, <= %6d differ, max rel deviation %.2e" % ((k,) + bad[k]))
    print("%d of %d variables differ between replays" % (len(bad), len(eng.params.manifest)))


if __name__ == "__main__":
    main()
