"""Round-5 correlation microbenchmark: the backward front end of a pyramid level -- global-atomic form (mh_tune_corr_row 0), row-owned scatter form (3), row-owned
gather form with staged operands (1, the default) and that launch without its gather part (5: timing) -- at the four MADNet level shapes (B = 1: what the step
runs) and at the SURVEY 8(d) protocol shape (B = 64); the 81-shift volume forward / backward (bf16 vs exact fp32, with and without the XCD-aware order).
Every timing = N launches recorded into ONE hipGraph and replayed (HIP events around the replays): a Python launch loop costs more than these kernels.
    python scripts/exp/mb_corr.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")):
    sys.path.insert(0, p)
import torch
from madnet_hip import _ffi, ops, benchtools as BT
from madnet_hip.plan import Recorder

lib = _ffi.lib()
st = torch.cuda.Stream()
sh = st.cuda_stream
dev = "cuda"


def graph_us(record, n=20, reps=10):
    """us per launch of `record(recorder)`: n copies in one captured graph"""
    rec = Recorder()
    for _ in range(n):
        record(rec)
    plan = rec.compile()
    with torch.cuda.stream(st):
        plan.run(lib, sh); st.synchronize()
        kname = lib.last_kernel().decode()
        plan.capture(lib, sh)
        ms = BT._time_ms(lib, st, lambda: plan.launch(lib, sh), reps)
        st.synchronize()
    lib.graph_destroy(plan.graph)
    return 1e3 * ms / n, kname


print("== backward front end of a level (mh_corr_warp_bwd: corr + concat gradient fused with the warp gradient), md 2; u in +-4 px and +-24 px")
for (B, H, W, C) in [(1, 96, 320, 32), (1, 48, 160, 64), (1, 24, 80, 96), (1, 12, 40, 128), (4, 96, 320, 32), (64, 96, 320, 32)]:
    D = 5
    ld = (C + D + 1 + 3) // 4 * 4
    L = torch.randn(B, H, W, C, device=dev); R = torch.randn(B, H, W, C, device=dev); g = torch.randn(B, H, W, ld, device=dev)
    dL = torch.zeros_like(L); dimg = torch.zeros_like(L); du = torch.zeros(B, H, W, device=dev)
    gv = ops.View(g, B, H, W, ld, ld)
    byts = float(B) * H * W * (8 * C + D + 3) * 4
    for amp in ((8.0, 48.0, -48.0) if B == 1 else (8.0,)):
        if amp > 0:
            u = (torch.rand(B, H, W, device=dev) - 0.5) * amp                    # white noise: adversarial (taps pile up at random)
        else:
            xx = torch.arange(W, device=dev, dtype=torch.float32)
            u = (0.5 * amp * torch.sin(xx / 25.0))[None, None, :].expand(B, H, W).contiguous() + (torch.rand(B, H, W, device=dev) - 0.5)      # smooth, like a disparity map
        Rw = torch.empty_like(R); ops.warp_fwd(lib, ops.view(R), u, ops.view(Rw)); torch.cuda.synchronize()
        for mode in (0, 1):
            lib.tune_corr_row(mode)
            us, k = graph_us(lambda r: ops.corr_warp_bwd(r, gv, ops.view(L), ops.view(Rw), ops.view(R), u, ops.view(dL), ops.view(dimg), du, 2, 1, coff=C, acc_l=True, copy_left=True),
                             n=(20 if B < 64 else 3), reps=(10 if B < 64 else 4))
            print("B=%-2d %3dx%3dx%3d |u|<=%2d%s %-16s %8.1f us  %7.0f GB/s algorithmic (%4.1f %% of 8 TB/s)   %s"
                  % (B, H, W, C, abs(amp) / 2, ("~" if amp < 0 else " "), {0: "atomic", 3: "row scatter", 1: "row gather", 5: "row, no gather"}[mode], us, byts / us / 1e3, byts / us / 1e3 / 80, k))
        lib.tune_corr_row(1)
    if B == 64:
        dR = torch.zeros_like(L)
        b2 = float(B) * H * W * (4 * C + D) * 4
        for direct in (0, 1):
            lib.tune_corr(direct)
            us, k = graph_us(lambda r: ops.corr_bwd(r, gv, ops.view(L), ops.view(R), ops.view(dL), ops.view(dR), 2, 1, coff=C, precision=0), n=3, reps=4)
            print("B=%-2d %3dx%3dx%3d  corr_bwd plain %8.1f us  %7.0f GB/s (%4.1f %%)   %s" % (B, H, W, C, us, b2 / us / 1e3, b2 / us / 1e3 / 80, k))
        lib.tune_corr(1)
    del L, R, g, Rw, dL, dimg

if "--no-d81" not in sys.argv:
    print("== 81-shift volume (md 40, C 128) at 96x320")
    for B in (16, 1):
        H, W, C, md = 96, 320, 128, 40
        D = 81
        L = torch.randn(B, H, W, C, device=dev); R = torch.randn(B, H, W, C, device=dev)
        vol = torch.empty(B, H, W, D, device=dev)
        ld = 84
        g = torch.randn(B, H, W, ld, device=dev); gv = ops.View(g, B, H, W, D, ld)
        dL = torch.empty_like(L); dR = torch.empty_like(R)
        bf, bb = float(B) * H * W * (2 * C + D) * 4, float(B) * H * W * (4 * C + D) * 4
        n = 3 if B == 16 else 20
        for tune, tag in ((1, "xcd order"), (5, "xcd, 2 grids"), (3, "plain order")):
            lib.tune_corr(tune)
            us, k = graph_us(lambda r: ops.corr_fwd(r, ops.view(L), ops.view(R), ops.view(vol), md, precision=1), n=n, reps=5)
            print("B=%-2d fwd bf16  %-12s %8.1f us  %7.0f GB/s (%4.1f %%)   %s" % (B, tag, us, bf / us / 1e3, bf / us / 1e3 / 80, k))
            us, k = graph_us(lambda r: ops.corr_bwd(r, gv, ops.view(L), ops.view(R), ops.view(dL), ops.view(dR), md, 1, coff=0, precision=1), n=n, reps=5)
            print("B=%-2d bwd bf16  %-12s %8.1f us  %7.0f GB/s (%4.1f %%)   %s" % (B, tag, us, bb / us / 1e3, bb / us / 1e3 / 80, k))
        lib.tune_corr(1)
        us, k = graph_us(lambda r: ops.corr_bwd(r, gv, ops.view(L), ops.view(R), ops.view(dL), ops.view(dR), md, 1, coff=0, precision=0), n=n, reps=3)
        print("B=%-2d bwd fp32               %8.1f us  %7.0f GB/s (%4.1f %%)   %s" % (B, us, bb / us / 1e3, bb / us / 1e3 / 80, k))
