"""Roofline micro-measurements used by bench.py (HIP events on the launch stream).

Peaks (MI355X_MICROARCH.md): fp32 MFMA 157.3 TFLOP/s dense; HBM3E 8.0 TB/s spec."""
import ctypes as C
import os
import torch

from . import ops

PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0          # dense (the 5 PF headline includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0


def _time_ms(lib, stream, fn, reps):
    s = C.c_void_p(stream.cuda_stream)
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
    for _ in range(3):
        fn()
    lib.stream_sync(s)
    lib.event_record(e0, s)
    for _ in range(reps):
        fn()
    lib.event_record(e1, s)
    ms = C.c_float()
    lib.event_elapsed_ms(e0, e1, C.byref(ms))
    lib.event_destroy(e0); lib.event_destroy(e1)
    return ms.value / reps


def _pmc_traffic(key):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r02_pmc_roofline.json, produced by
    scripts/gpu_pmc_r02.sh + scripts/pmc_summarize_r02.py); None if the file or the key is absent."""
    import json
    import os
    f = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "profiles", "r02_pmc_roofline.json")
    try:
        return json.load(open(f))[key]["traffic_bytes"]
    except Exception:
        return None


def roofline(lib, eng, stream, reps=20):
    """Dominant forward kernel of the step = the 128->128 3x3 conv at 1/4 resolution (context-2/3, G2 disp-2: 27 of the 70
    forward GFLOP).  achieved = algorithmic flops (2*Ho*Wo*9*Cin*Cout) / mean launch time over `reps` launches on the bench
    stream, using the engine's own buffers and weights (context-2: dilation 2), in the engine's FORWARD arithmetic.  The kernel
    name is what the dispatcher actually launched (mh_last_kernel), not a hard-coded string."""
    from . import engine as E
    sh = stream.cuda_stream
    x = ops.view(eng.Cx[0]); o = ops.view(eng.Cx[1])
    w = eng.W_(E.ctx_name(2)); b = eng.b_(E.ctx_name(2))
    fwd_code, bwd_code = ops.PRECISION_CODES[eng.precision]
    wb = eng.Wb_(E.ctx_name(2)) if hasattr(eng, "Wb_") else None       # the layer's fragment bank (packed by the step's own mh_pack_weights launch)
    flops = 2.0 * x.B * x.H * x.W * 9 * 128 * 128
    algo_bytes = 4.0 * (2 * x.B * x.H * x.W * 128 + 9 * 128 * 128)

    def entry(code, fn, what, pmc_key):
        ms = _time_ms(lib, stream, fn, reps)
        kname = lib.last_kernel().decode()
        # every product = 1 bf16 MFMA (code 1), 3 bf16 MFMAs (code 2) or fp32 MFMA steps (code 0): the roofline is priced on the
        # ALGORITHMIC flops against the peak of the instruction that runs (dense bf16 2.5 PF for codes 1 and 2)
        peak = PEAK_F32_MFMA_TFLOPS if (code == 0 or "f32" in kname.split("tile")[0]) else PEAK_BF16_MFMA_TFLOPS
        ach = flops / (ms * 1e-3) / 1e12
        tr = _pmc_traffic(pmc_key)
        x3 = code == 2 and peak == PEAK_BF16_MFMA_TFLOPS
        if x3:
            # split-bf16 issues 3 bf16 MFMAs per algorithmic product: the ceiling of ALGORITHMIC flops is a third of the dense bf16 peak
            peak = PEAK_BF16_MFMA_TFLOPS / 3.0
        return {"kernel": kname, "op": what, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "peak_note": ("dense bf16 MFMA peak / 3 (three MFMAs per product); against the plain 2500 TFLOP/s the algorithmic rate is %.3f" % (ach / PEAK_BF16_MFMA_TFLOPS)) if x3 else None,
                "arithmetic": {0: "f32 MFMA", 1: "bf16 MFMA, f32 accumulate", 2: "split-bf16: 3 bf16 MFMAs per product (mfma issue rate = 3x achieved), f32 accumulate"}[code],
                "traffic": tr, "traffic_source": ("profiles/r02_pmc_roofline.json (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/gpu_pmc_r02.sh, key %s; not re-measured in this run)" % pmc_key) if tr is not None else None,
                "launch_ms": ms, "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": algo_bytes}

    def conv_fwd(code):
        def fn():
            ops.PRECISION = code
            try:
                ops.conv2d_fwd(lib, x, w, b, o, dil=2, alpha=E.ALPHA, stream=sh, wb=(wb if code == 2 else None))
            finally:
                ops.PRECISION = 0
        return fn

    rl = entry(fwd_code, conv_fwd(fwd_code), "forward 3x3 128->128 @ %dx%d dil 2 (context-2)" % (x.H, x.W),
               {0: "none", 1: "conv_fwd_bf16_patch_3x3_128_128_96x320",
                2: "conv_fwd_x3_bank_3x3_128_128_96x320" if wb is not None else "conv_fwd_x3_patch_3x3_128_128_96x320"}[fwd_code])
    extra = {}
    # the same layer's input gradient and filter gradient in the BACKWARD arithmetic: by time the filter gradients are the
    # largest kernel family of the step (22 launches x 17 us)
    try:
        dz = ops.view(eng.dCx[1]); dx = ops.view(eng.dCx[0])

        def dgrad():
            ops.PRECISION = bwd_code
            try:
                ops.conv2d_dgrad(lib, dz, w, dx, dil=2, mask_ref=x, mask_alpha=E.ALPHA, stream=sh)
            finally:
                ops.PRECISION = 0
        extra["roofline_dgrad"] = entry(bwd_code, dgrad, "input gradient of the same layer", "conv_dgrad_bf16_patch_3x3_128_128_96x320" if bwd_code == 1 else "none")
        dw = torch.empty_like(w); db = torch.zeros(128, device=eng.dev)
        wsa = ops.WgradWorkspace(eng.dev)
        segs, keep = [], []
        ops.PRECISION = bwd_code
        try:
            ops.conv2d_wgrad_partial(lib, lib, wsa, segs, x, dz, dw, db, dil=2, stream=sh)     # sizes the workspace once
        finally:
            ops.PRECISION = 0

        def wgrad():
            ops.PRECISION = bwd_code
            try:
                wsa.reset(); s2 = []
                ops.conv2d_wgrad_partial(lib, lib, wsa, s2, x, dz, dw, db, dil=2, stream=sh)
            finally:
                ops.PRECISION = 0
        extra["roofline_wgrad"] = entry(bwd_code, wgrad, "filter gradient of the same layer (partial sums only; the split reduction is one launch per batch of layers)",
                                        "wgrad_bf16_partial_3x3_128_128_96x320" if bwd_code == 1 else "none")
        extra["roofline_wgrad"]["splits"] = segs[0][3] if segs else 1
        extra["roofline_wgrad"]["workspace_bytes_per_launch"] = 4.0 * (segs[0][2] * segs[0][3] if segs else 0)
    except Exception as ex:
        extra["roofline_wgrad"] = {"error": repr(ex)}
    # correlation protocol: level-2 shape with B=64 streams (working set > 256 MiB
    # Infinity Cache) for the HBM claim, plus the in-situ B=1 time (cache resident).
    try:
        dev = eng.dev
        Bc, H, W, Cc, md = 64, x.H, x.W, 32, eng.md
        L = torch.randn(Bc, H, W, Cc, device=dev); R = torch.randn(Bc, H, W, Cc, device=dev)
        D = 2 * md + 1
        out = torch.empty(Bc, H, W, D, device=dev)
        ms_c = _time_ms(lib, stream, lambda: ops.corr_fwd(lib, ops.view(L), ops.view(R), ops.view(out), md, stream=sh), 10)
        byts = float(Bc) * H * W * (2 * Cc + D) * 4
        g = byts / (ms_c * 1e-3) / 1e9
        extra["roofline_corr"] = {"kernel": lib.last_kernel().decode() + " (B=64 x %dx%dx%d, D=%d)" % (H, W, Cc, D), "bound": "hbm",
                                  "achieved": g, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": g / PEAK_HBM_GBS,
                                  "traffic": _pmc_traffic("corr_fwd_B64_96x320x32_D5"), "traffic_source": "profiles/r02_pmc_roofline.json (static: rocprofv3 --pmc passes of scripts/gpu_pmc_r02.sh; not re-measured in this run)", "launch_ms": ms_c, "algorithmic_bytes_per_launch": byts}
        L1, R1 = L[:1].contiguous(), R[:1].contiguous(); o1 = out[:1].contiguous()
        ms_1 = _time_ms(lib, stream, lambda: ops.corr_fwd(lib, ops.view(L1), ops.view(R1), ops.view(o1), md, stream=sh), 20)
        extra["roofline_corr"]["in_situ_B1_ms"] = ms_1
        extra["roofline_corr"]["in_situ_note"] = "B=1 (what the step runs) is cache resident and launch bound: %.1f GB/s algorithmic" % (float(H * W * (2 * Cc + D) * 4) / (ms_1 * 1e-3) / 1e9)
    except Exception as ex:       # never let the auxiliary measurement kill the bench line
        extra["roofline_corr"] = {"error": str(ex)}
    return rl, extra
