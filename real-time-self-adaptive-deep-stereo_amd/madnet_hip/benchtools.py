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
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r01_pmc_roofline.json,
    produced by scripts/gpu_pmc.sh); None if the file is absent."""
    import json
    import os
    f = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "profiles", "r01_pmc_roofline.json")
    try:
        return json.load(open(f))[key]["traffic_bytes"]
    except Exception:
        return None


def roofline(lib, eng, stream, reps=20):
    """Dominant kernel of the step = the 128->128 3x3 implicit-GEMM conv at 1/4 resolution
    (context-2/3, G2 disp-2: 27 of the 70 forward GFLOP; its dgrad is the same kernel).
    achieved = algorithmic flops (2*Ho*Wo*9*Cin*Cout) / mean launch time over `reps` launches on
    the bench stream, using the engine's own buffers and weights (context-2: dilation 2)."""
    from . import engine as E
    sh = stream.cuda_stream
    x = ops.view(eng.Cx[0]); o = ops.view(eng.Cx[1])
    w = eng.W_(E.ctx_name(2)); b = eng.b_(E.ctx_name(2))

    prec = ops.PRECISION_CODES[eng.precision] if hasattr(ops, "PRECISION_CODES") else (1 if eng.precision == "bf16" else 0)
    peak = PEAK_BF16_MFMA_TFLOPS if prec == 1 else PEAK_F32_MFMA_TFLOPS

    def conv():
        ops.PRECISION = prec
        try:
            ops.conv2d_fwd(lib, x, w, b, o, dil=2, alpha=E.ALPHA, stream=sh)
        finally:
            ops.PRECISION = 0

    ms = _time_ms(lib, stream, conv, reps)
    flops = 2.0 * x.B * x.H * x.W * 9 * 128 * 128
    ach = flops / (ms * 1e-3) / 1e12
    kname = ("conv_patch_kernel<4,2,2,4,fwd,K=128> (patch-staged bf16, 3x3 128->128 @ %dx%d, dil 2)" if prec == 1 and os.environ.get("MH_CONV_PATCH", "1") != "0"
             else "conv_igemm_kernel (3x3 128->128 @ %dx%d, dil 2; tile chosen by conv_dispatch)") % (x.H, x.W)
    rl = {"kernel": kname,
          "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
          "frac": ach / peak, "arithmetic": "bf16 MFMA, f32 accumulate" if prec == 1 else "f32 MFMA",
          "traffic": _pmc_traffic("conv_3x3_128_128_96x320_bf16" if prec == 1 else "conv_3x3_128_128_96x320"),
          "launch_ms": ms, "algorithmic_flops_per_launch": flops}
    # correlation protocol: level-2 shape with B=64 streams (working set > 256 MiB
    # Infinity Cache) for the HBM claim, plus the in-situ B=1 time (cache resident).
    extra = {}
    try:
        dev = eng.dev
        Bc, H, W, Cc, md = 64, x.H, x.W, 32, eng.md
        L = torch.randn(Bc, H, W, Cc, device=dev); R = torch.randn(Bc, H, W, Cc, device=dev)
        D = 2 * md + 1
        out = torch.empty(Bc, H, W, D, device=dev)
        ms_c = _time_ms(lib, stream, lambda: ops.corr_fwd(lib, ops.view(L), ops.view(R), ops.view(out), md, stream=sh), 10)
        byts = float(Bc) * H * W * (2 * Cc + D) * 4
        g = byts / (ms_c * 1e-3) / 1e9
        extra["roofline_corr"] = {"kernel": "corr_fwd_direct<8,5> (B=64 x %dx%dx%d, D=%d)" % (H, W, Cc, D), "bound": "hbm",
                                  "achieved": g, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": g / PEAK_HBM_GBS,
                                  "traffic": _pmc_traffic("corr_fwd_B64_96x320x32_D5"), "launch_ms": ms_c, "algorithmic_bytes_per_launch": byts}
        L1, R1 = L[:1].contiguous(), R[:1].contiguous(); o1 = out[:1].contiguous()
        ms_1 = _time_ms(lib, stream, lambda: ops.corr_fwd(lib, ops.view(L1), ops.view(R1), ops.view(o1), md, stream=sh), 20)
        extra["roofline_corr"]["in_situ_B1_ms"] = ms_1
    except Exception as ex:       # never let the auxiliary measurement kill the bench line
        extra["roofline_corr"] = {"error": str(ex)}
    return rl, extra
