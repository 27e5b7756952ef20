"""Roofline micro-measurements used by bench.py (HIP events on the launch stream).

Peaks (MI355X_MICROARCH.md): fp32 MFMA 157.3 TFLOP/s dense; HBM3E 8.0 TB/s spec."""
import ctypes as C
import os
import torch

from . import ops

PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0          # dense (the 5 PF headline includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0


WARMUP_LAUNCHES = 3         # untimed launches in front of every timed burst
TABLE_REPS = 10             # timed launches per op of a plan table


def _time_ms(lib, stream, fn, reps):
    s = C.c_void_p(stream.cuda_stream)
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
    for _ in range(WARMUP_LAUNCHES):
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


PMC_FILES = ("r06_pmc_roofline.json", "r05_pmc_roofline.json")        # the first that exists is THE source (one file: VERDICT r04 weak 9b)
PMC_SOURCE = None           # the file the last _pmc_traffic() hit came from


_PMC_CACHE = None


def _pmc_json():
    """(file name, parsed JSON) of THE committed PMC summary: the first of PMC_FILES that exists (parsed once per process)"""
    global _PMC_CACHE
    if _PMC_CACHE is None:
        import json
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "profiles")
        _PMC_CACHE = (None, {})
        for f in PMC_FILES:
            try:
                _PMC_CACHE = (f, json.load(open(os.path.join(d, f))))
                break
            except Exception:
                continue
    return _PMC_CACHE


def _pmc_traffic(key):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r04_pmc_roofline.json, produced by scripts/gpu_pmc.sh +
    scripts/pmc_summarize.py; the previous round's file as a fallback); `key` = a kernel string as mh_last_kernel reports it, or the name of a fixed
    roofline entry (roofline_fwd, roofline_corr, ...: resolved through the file's fixed_kernels map).  None if absent."""
    global PMC_SOURCE
    f, j = _pmc_json()
    try:
        if key not in j and key in j.get("fixed_kernels", {}):
            key = j["fixed_kernels"][key]              # roofline_* name -> the kernel string it ran
        v = j[key]["traffic_bytes"]
        PMC_SOURCE = "profiles/" + f
        return v
    except Exception:
        return None


def _pmc_entry(key):
    f, j = _pmc_json()
    if key not in j and key in j.get("fixed_kernels", {}):
        key = j["fixed_kernels"][key]
    return j.get(key)


def _fixed_source(key):
    """provenance string of a fixed roofline entry's `traffic` (ONE file: the first of PMC_FILES that exists), None when the file has no such entry"""
    if _pmc_traffic(key) is None:
        return None
    return "%s, fixed_kernels.%s: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/gpu_pmc.sh (L2 flushed before the measured launch); collected beside this round's committed bench line, not inside this process" % (PMC_SOURCE, key)


def family_key(kernel):
    """kernel string -> family: the template name and its arguments, except for the planes kernel whose tile instances are ONE family per
    arithmetic (the split-bf16 forward layers / their plain-bf16 input gradients)"""
    import re
    if kernel.startswith("conv_planes_kernel<"):
        return "conv_planes_kernel<dgrad,bf16>" if "<dgrad," in kernel else "conv_planes_kernel<fwd,bf16x3>"
    return re.split(r" tile | layers | grid | K=| \(", kernel)[0]


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
        tr = _pmc_traffic(kname)
        if tr is None:
            tr = _pmc_traffic(pmc_key)
        x3 = code == 2 and peak == PEAK_BF16_MFMA_TFLOPS
        # frac = ALGORITHMIC flops / time / the dense peak of the instruction family (2.5 PF for bf16 MFMA, whatever the number of
        # MFMAs a product costs); the share of the MFMA ISSUE rate the kernel sustains (3 instructions per product in split-bf16) is reported beside it
        return {"kernel": kname, "op": what, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "mfma_issue_frac": (3.0 if x3 else 1.0) * ach / peak,
                "arithmetic": {0: "f32 MFMA", 1: "bf16 MFMA, f32 accumulate", 2: "split-bf16: 3 bf16 MFMAs per product (mfma_issue_frac = 3 x frac), f32 accumulate"}[code],
                "traffic": tr, "traffic_source": ("%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/gpu_pmc.sh on the same kernels and shapes, "
                                                  "key %s; collected in the run that produced this round's committed bench line, not inside this process)" % (PMC_SOURCE, pmc_key)) if tr is not None else None,
                "launch_ms": ms, "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": algo_bytes}

    planes = getattr(eng, "use_planes", False) and E.ctx_name(2) in getattr(eng, "banks32", {})
    keep = []
    if planes:
        # what the plan runs: the layer from the hi / lo planes of its input (written by context-1's epilogue in the step; split here, outside the
        # timed launches), results as planes only (the plan's post-pass elides the fp32 store of this layer)
        _, xp = eng._planes_of(x)
        _, op_ = eng._planes_of(o)
        ops.plane_split(lib, [(x, xp)], eng.dev, keep, stream=sh)
        algo_bytes = 2.0 * 2 * (2 * x.B * x.H * x.W * 128 + 9 * 128 * 128)

    def conv_fwd(code):
        def fn():
            if planes and code == 2:
                ops.conv2d_planes(lib, xp, w, eng.banks32[E.ctx_name(2)], b, out=None, out_planes=op_, dil=2, alpha=E.ALPHA, stream=sh)
                return
            ops.PRECISION = code
            try:
                ops.conv2d_fwd(lib, x, w, b, o, dil=2, alpha=E.ALPHA, stream=sh, wb=(wb if code == 2 else None))
            finally:
                ops.PRECISION = 0
        return fn

    rl = entry(fwd_code, conv_fwd(fwd_code), "forward 3x3 128->128 @ %dx%d dil 2 (context-2)%s" % (x.H, x.W, " from hi / lo planes, planes out: what the plan runs" if planes else ""),
               "roofline_fwd")
    extra = {}
    # the same layer's input gradient and filter gradient in the BACKWARD arithmetic: by time the filter gradients are the
    # largest kernel family of the step (22 launches x 17 us)
    try:
        dz = ops.view(eng.dCx[1]); dx = ops.view(eng.dCx[0])

        bwd_planes = planes and bwd_code == 1 and E.ctx_name(2) in getattr(eng, "banks32t", {})
        if bwd_planes:
            # what the plan runs: dz from its shadow, the mask from the activation's hi plane, the result as a shadow only
            key = (dz.ptr, dz.B, dz.H, dz.W, dz.C)
            dzs = eng.shadows.get(key) or ops.Shadow(dz.B, dz.H, dz.W, dz.C, eng.dev)
            key = (dx.ptr, dx.B, dx.H, dx.W, dx.C)
            dxs = eng.shadows.get(key) or ops.Shadow(dx.B, dx.H, dx.W, dx.C, eng.dev)
            ops.shadow_cast(lib, [(dz, dzs)], eng.dev, keep, stream=sh)

        def dgrad():
            if bwd_planes:
                ops.conv2d_planes_bwd(lib, dzs, w, eng.banks32t[E.ctx_name(2)], dx=None, dx_shadow=dxs, mask_shadow=xp.hi, mask_alpha=E.ALPHA, dil=2, stream=sh)
                return
            ops.PRECISION = bwd_code
            try:
                ops.conv2d_dgrad(lib, dz, w, dx, dil=2, mask_ref=x, mask_alpha=E.ALPHA, stream=sh)
            finally:
                ops.PRECISION = 0
        extra["roofline_dgrad"] = entry(bwd_code, dgrad, "input gradient of the same layer" + (" (bf16 shadows in / out: what the plan runs)" if bwd_planes else ""),
                                        "roofline_dgrad")
        if bwd_planes:
            extra["roofline_dgrad"]["algorithmic_bytes_per_launch"] = 2.0 * (2 * x.B * x.H * x.W * 128 + 9 * 128 * 128)
        dw = torch.empty_like(w); db = torch.zeros(128, device=eng.dev)
        wsa = ops.WgradWorkspace(eng.dev)
        keep = []
        if bwd_code == 1 and getattr(eng, "use_stream", False):
            # what the step runs: bf16 shadows (written by the producers' epilogues there; cast here, outside the timed launches) -> mh_wgrad_stream
            def stream_entry(layers, what, pmc_key, nwaves):
                items, pairs, fl = [], [], 0.0
                for (xv, zv, dwt, dbt, dil) in layers:
                    xs, zs = ops.Shadow(xv.B, xv.H, xv.W, xv.C, eng.dev), ops.Shadow(zv.B, zv.H, zv.W, zv.C, eng.dev)
                    pairs += [(xv, xs), (zv, zs)]
                    items.append((xs, zs, dwt, dbt, dil))
                    fl += 2.0 * xv.B * xv.H * xv.W * 9 * xv.C * zv.C
                ops.shadow_cast(lib, pairs, eng.dev, keep, stream=sh)
                from .plan import Recorder
                rec = Recorder(); segs = []
                ops.wgrad_stream(rec, lib, wsa, segs, items, eng.dev, keep, nwaves=nwaves)
                pl = rec.compile()
                ms = _time_ms(lib, stream, lambda: pl.run(lib, sh), reps)
                pl.run(lib, sh); kname = lib.last_kernel().decode()
                ach = fl / (ms * 1e-3) / 1e12
                tr = _pmc_traffic(kname)
                if tr is None:
                    tr = _pmc_traffic(pmc_key)
                byts = sum(2.0 * xv.B * xv.H * xv.W * (ops.shadow_ld(xv.C) + ops.shadow_ld(zv.C)) + 4.0 * 9 * xv.C * zv.C for xv, zv, _, _, _ in layers)
                return {"kernel": kname, "op": what, "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / PEAK_BF16_MFMA_TFLOPS, "mfma_issue_frac": ach / PEAK_BF16_MFMA_TFLOPS, "arithmetic": "bf16 MFMA (32x32x16), f32 accumulate; operands = bf16 shadows",
                        "traffic": tr, "traffic_source": ("%s key %s (rocprofv3 --pmc passes of scripts/gpu_pmc.sh)" % (PMC_SOURCE, pmc_key)) if tr is not None else None,
                        "launch_ms": ms, "algorithmic_flops_per_launch": fl, "algorithmic_bytes_per_launch": byts,
                        "splits": [sg[3] for sg in segs if len(sg) == 4], "workspace_bytes_per_launch": 4.0 * sum(sg[2] * sg[3] for sg in segs)}
            nw = 4 if x.B == 1 else 8
            extra["roofline_wgrad"] = stream_entry([(x, dz, dw, db, 2)], "filter gradient of the same layer, alone in its launch (streaming kernel; partial sums only)",
                                                   "roofline_wgrad", nw)
            k = 2
            h, wd = eng.fshape[E.FEAT[k]][0], eng.fshape[E.FEAT[k]][1]
            cin = eng.fshape[E.FEAT[k]][2] + eng.D + 1
            lay = []
            for j in range(1, 7):
                xin = ops.View(eng.dsi[k], x.B, h, wd, cin, eng.dsi_ld[k]) if j == 1 else ops.view(eng.E[k][j - 2])
                dzz = ops.view(eng.dV[k]) if j == 6 else ops.view(eng.dE[k][j - 1])
                lay.append((xin, dzz, torch.empty_like(eng.W_(E.est_name(k, j))), torch.zeros(dzz.C, device=eng.dev), 1))
            extra["roofline_wgrad_batch"] = stream_entry(lay, "filter gradients of the six estimator-2 layers in ONE launch (what the step runs per backward batch)",
                                                         "roofline_wgrad_batch", nw)
        else:
            segs = []
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
                                  "traffic": _pmc_traffic("roofline_corr"), "traffic_source": _fixed_source("roofline_corr"), "launch_ms": ms_c, "algorithmic_bytes_per_launch": byts}
        L1, R1 = L[:1].contiguous(), R[:1].contiguous(); o1 = out[:1].contiguous()
        ms_1 = _time_ms(lib, stream, lambda: ops.corr_fwd(lib, ops.view(L1), ops.view(R1), ops.view(o1), md, stream=sh), 20)
        extra["roofline_corr"]["in_situ_B1_ms"] = ms_1
        extra["roofline_corr"]["in_situ_note"] = "B=1 (what the step runs) is cache resident and launch bound: %.1f GB/s algorithmic" % (float(H * W * (2 * Cc + D) * 4) / (ms_1 * 1e-3) / 1e9)
    except Exception as ex:       # never let the auxiliary measurement kill the bench line
        extra["roofline_corr"] = {"error": str(ex)}
    return rl, extra


# plan op kind -> the kernel(s) behind it, for the kinds whose entry point does not report an instance through mh_last_kernel (csrc/lib.hip: mh_plan_run)
OP_KERNEL_NAMES = {
    "OP_WARP_FWD": "warp_fwd_kernel", "OP_WARP_BWD": "warp_bwd_kernel", "OP_RESIZE_FWD": "resize_fwd_kernel (legacy bilinear + scale / relu / crop)",
    "OP_RESIZE_BWD": "resize_bwd_kernel (gradient of the final / _make_disp resize)", "OP_PAD_REFLECT": "pad_reflect_kernel (pad_image + cast, both frames)",
    "OP_LOSS": "loss_tile_kernel + loss_final_kernel (reprojection loss SSIM + L1, value and d/d disparity)", "OP_METRICS": "metrics_kernel + metrics_final_kernel (EPE / bad3)",
    "OP_MOMENTUM": "momentum_kernel (optimizer)", "OP_COPY_CH": "copy_channels_kernel", "OP_LEAKY_BWD": "leaky_bwd_kernel", "OP_FILL": "fill_kernel (zero of gradient ranges)",
    "OP_BIAS_GRAD": "bias_grad_kernel", "OP_WGRAD_REDUCE": "wgrad_reduce_kernel (sum over the filter-gradient splits of a batch)", "OP_PROXY_LOSS": "proxy_loss kernels",
    "OP_SUPERVISED_LOSS": "supervised_loss kernels", "OP_ADAM": "adam_kernel", "OP_ADAM_ADVANCE": "adam_advance_kernel", "OP_RESIZE_IMAGE": "resize_image_fwd_kernel",
    "OP_PACK_W": "pack_weights_kernel (MFMA fragment banks of every layer, once per step)", "OP_SHADOW_CAST": "shadow_cast_kernel (bf16 shadows of a filter-gradient batch)",
    "OP_HEAD_BWD": "head_bwd_kernel (disparity head: output gradient + 3x3 Cin->1 input gradient)", "OP_HEAD_FWD": "conv_n1_fwd_kernel (disparity head with extra destinations)",
    "OP_PLANE_SPLIT": "plane_split_kernel (hi / lo bf16 planes + fused concat)", "OP_STAMP": "stamp_kernel", "OP_DET_FLUSH": "det_flush_kernel",
}
_REPORTING = ("OP_CONV", "OP_WGRAD", "OP_WGRAD_PARTIAL", "OP_WGRAD_STREAM", "OP_CORR_FWD", "OP_CORR_BWD", "OP_LEVEL_FRONT", "OP_CORR_WARP_BWD", "OP_CONV_PLANES", "OP_CONV_IMAGE",
              "OP_CONV_PLANES_BWD")


def op_kernel_name(lib, op):
    """what ran for a plan op: the instance string the dispatcher noted (mh_last_kernel) or, for the single-kernel entry points, the static name"""
    from . import _ffi
    reporting = tuple(getattr(_ffi, n) for n in _REPORTING)
    if op.kind in reporting:
        k = lib.last_kernel().decode()
        if op.kind == _ffi.OP_CONV_PLANES_BWD:
            k = k.replace("conv_planes_kernel<", "conv_planes_kernel<dgrad,")
        return k
    for n, label in OP_KERNEL_NAMES.items():
        if getattr(_ffi, n, None) == op.kind:
            return label
    return "op kind %d" % op.kind


def plan_table(lib, plan, stream, reps=None):
    """Every op of a recorded plan timed ALONE (HIP events on the launch stream, `reps` launches each) with the kernel the dispatcher chose
    (mh_last_kernel): which kernel family the step spends its time in, from the plan's own launch table.  The ops mutate the engine they were
    recorded on (optimizer, accumulating gradients): run it on a scratch engine.  Returns (rows, families): rows = [(index, kind, kernel, us)],
    families = {kernel template: {"launches", "us_per_step", "top_us", "top_index"}} sorted by time."""
    from . import _ffi
    rows = []
    sh = stream.cuda_stream
    reps = reps or TABLE_REPS
    for i in range(plan.n):
        one = (_ffi.Op * 1)(plan.arr[i])
        one[0].i[26] = 0                                    # on the caller's stream, no join
        us = 1e3 * _time_ms(lib, stream, lambda: lib.plan_run(one, 1, C.c_void_p(sh)), reps)
        rows.append((i, int(plan.arr[i].kind), op_kernel_name(lib, plan.arr[i]), us))
    fam = {}
    for i, kind, k, us in rows:
        key = family_key(k)
        f = fam.setdefault(key, {"launches": 0, "us_per_step": 0.0, "top_us": 0.0, "top_index": -1})
        f["launches"] += 1; f["us_per_step"] += us
        if us > f["top_us"]:
            f["top_us"], f["top_index"] = us, i
    return rows, dict(sorted(fam.items(), key=lambda kv: -kv[1]["us_per_step"]))


def _peak_of(kernel):
    return PEAK_F32_MFMA_TFLOPS if (",f32," in kernel.replace(" ", "") or "wgrad_kernel<" in kernel) else PEAK_BF16_MFMA_TFLOPS


def family_report(lib, plans, stream, weights=None, nfam=12):
    """Dominant kernel family + family table of a step, from the launch table(s) of its recorded plan(s) (plan_table: every op timed alone, HIP events).
    plans: [Plan]; weights: how often each plan runs per step on average (MAD: the share of the steps that sampled the block; default 1 each).
    Returns {"roofline", "kernel_families", "kernel_time_sum_us", "tables"}: `roofline` prices the family the step spends most time in -- algorithmic
    flops of its launches / their summed time against the dense MFMA peak; an HBM-bound top family (no flops) is priced in bytes."""
    weights = list(weights) if weights is not None else [1.0] * len(plans)
    fam, tables, tot = {}, [], 0.0
    for pl, wgt in zip(plans, weights):
        if wgt <= 0:
            tables.append(None)
            continue
        rows, _ = plan_table(lib, pl, stream)
        tables.append(rows)
        for i, kind, k, us in rows:
            fl, by = pl.work.get(i, op_work(pl.arr[i]))
            f = fam.setdefault(family_key(k), {"launches": 0.0, "us": 0.0, "flops": 0.0, "bytes": 0.0, "traffic": 0.0, "covered": True, "top_us": 0.0, "top": None})
            f["launches"] += wgt; f["us"] += wgt * us; f["flops"] += wgt * fl; f["bytes"] += wgt * by
            tr = _pmc_traffic(k)
            if tr is None:
                f["covered"] = False
            else:
                f["traffic"] += wgt * tr
            if us > f["top_us"]:
                f["top_us"], f["top"] = us, {"kernel": k, "launch_ms": us * 1e-3, "flops": fl, "bytes": by, "traffic": tr}
            tot += wgt * us
    fam = dict(sorted(fam.items(), key=lambda kv: -kv[1]["us"]))
    kf = []
    for k, v in list(fam.items())[:nfam]:
        ent = {"kernel": k, "launches": v["launches"], "us_per_step": v["us"]}
        if v["flops"] > 0 and v["us"] > 0:
            ent["achieved_tflops"] = v["flops"] / (v["us"] * 1e-6) / 1e12
            ent["frac"] = ent["achieved_tflops"] / _peak_of(k)
        if v["bytes"] > 0 and v["us"] > 0:
            ent["algorithmic_gbs"] = v["bytes"] / (v["us"] * 1e-6) / 1e9
            ent["hbm_frac"] = ent["algorithmic_gbs"] / PEAK_HBM_GBS
            ent["algorithmic_bytes_per_step"] = v["bytes"]
        ent["traffic"] = v["traffic"] if (v["covered"] and v["traffic"] > 0) else None
        kf.append(ent)
    name, f = next(iter(fam.items()))
    top = f["top"]
    src = ("%s: sum over this family's launches, keyed by their kernel strings (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same plans)" % PMC_SOURCE)
    if f["flops"] > 0:
        peak = _peak_of(top["kernel"])
        ach = f["flops"] / (f["us"] * 1e-6) / 1e12
        x3 = 3.0 if "bf16x3" in top["kernel"] or "bf16x3" in name else 1.0
        rl = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "mfma_issue_frac": x3 * ach / peak}
        top_ach = top["flops"] / (top["launch_ms"] * 1e-3) / 1e12 if top["launch_ms"] > 0 else None
        rl["longest_launch"] = {"kernel": top["kernel"], "launch_ms": top["launch_ms"], "achieved": top_ach, "frac": (top_ach / peak) if top_ach else None, "traffic": top["traffic"]}
    else:
        ach = f["bytes"] / (f["us"] * 1e-6) / 1e9 if f["us"] > 0 else 0.0
        rl = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS}
    covered = f["covered"] and f["traffic"] > 0
    rl.update({"traffic": f["traffic"] if covered else None, "traffic_source": src if covered else None,
               "launch_ms": f["us"] * 1e-3 / f["launches"] if f["launches"] else None, "launches_per_step": f["launches"], "us_per_step": f["us"],
               "algorithmic_flops_per_step": f["flops"], "algorithmic_bytes_per_step": f["bytes"], "share_of_kernel_time": f["us"] / tot if tot else None,
               "selection": "the kernel family the recorded plan spends most time in, from the plan's own launch table (every op timed alone with HIP events, 10 launches "
                            "each); achieved = the family's algorithmic flops per step / its summed launch time; frac against the DENSE bf16 MFMA peak"})
    return {"roofline": rl, "kernel_families": kf, "kernel_time_sum_us": tot, "tables": tables}


def op_work(op):
    """(algorithmic flops, algorithmic bytes) of a conv / filter-gradient op record (SURVEY 8(d) definitions), else (0, 0)"""
    from . import _ffi
    if op.kind in (_ffi.OP_CONV_PLANES, _ffi.OP_CONV_PLANES_BWD):
        # planes layers: operands and results are bf16 planes: in 2 (hi, lo) x 2 B forward / 1 x 2 B backward per element, out likewise (+ 4 B where the
        # fp32 copy is stored)
        i = op.i
        B, H, W, K, N = i[0], i[1], i[2], i[5], i[6]
        fwd = op.kind == _ffi.OP_CONV_PLANES
        cin, cout = (K, N) if fwd else (N, K)
        pl = 2 if fwd else 1
        f32 = 4.0 if op.p[4 if fwd else 3] else 0.0
        mask = 2.0 * cout if (not fwd and op.p[2]) else 0.0      # the input gradient reads the activation's hi plane once for the sign test
        return 2.0 * B * H * W * 9 * K * N, B * H * W * (cin * 2.0 * pl + cout * (2.0 * pl + f32) + mask) + 9 * K * N * 2.0 * pl
    # correlation family: HBM-bound, bytes = every operand once (forward B H W (2C + D) 4, gradient B H W (4C + D) 4; the fused forms add what they fuse)
    if op.kind == _ffi.OP_CORR_FWD:
        B, H, W, Cc, md, st = op.i[4], op.i[5], op.i[6], op.i[7], op.i[8], op.i[9]
        D = 2 * md // max(st, 1) + 1
        return 2.0 * B * H * W * Cc * D, 4.0 * B * H * W * (2 * Cc + D + (Cc if op.i[10] else 0))
    if op.kind == _ffi.OP_CORR_BWD:
        B, H, W, Cc, md, st = op.i[9], op.i[10], op.i[11], op.i[12], op.i[13], op.i[14]
        D = 2 * md // max(st, 1) + 1
        return 4.0 * B * H * W * Cc * D, 4.0 * B * H * W * (4 * Cc + D + (Cc if op.i[15] else 0))
    if op.kind == _ffi.OP_CORR_WARP_BWD:
        B, H, W, Cc, md, st = op.i[8], op.i[9], op.i[10], op.i[11], op.i[12], op.i[13]
        D = 2 * md // max(st, 1) + 1
        return 4.0 * B * H * W * Cc * D, 4.0 * B * H * W * (8 * Cc + D + 3)        # reads g (C + D + 1), L, Rw, the right features, u, dL, dimg; writes dL, dimg, du
    if op.kind == _ffi.OP_CONV_IMAGE:
        NB, H0, W0, Cc, Hp, Wp, N, st = op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], op.i[8], op.i[9]
        Ho, Wo = (Hp + st - 1) // st, (Wp + st - 1) // st
        return 2.0 * NB * Ho * Wo * 9 * Cc * N, 4.0 * (NB * H0 * W0 * Cc + NB * Ho * Wo * N + 9 * Cc * N) + (2.0 * NB * Ho * Wo * N if op.p[4] else 0.0)
    if op.kind == _ffi.OP_LEVEL_FRONT:
        B, H, W, Cc, md = op.i[7], op.i[8], op.i[9], op.i[10], op.i[11]
        D = 2 * md + 1
        fl, by = 2.0 * B * H * W * Cc * D, 4.0 * B * H * W * (2 * Cc + (Cc + D + 1) + Cc + 1)        # reads L, R; writes [L | corr | u], the warped features, u
        if op.p[8]:                 # + the coarser level's disparity head (3x3, K -> 1): reads its input once, writes the coarse disparity
            Hc, Wc, K = op.i[0], op.i[1], op.i[15]
            fl += 2.0 * B * Hc * Wc * 9 * K
            by += 4.0 * B * Hc * Wc * (K + 1)
        return fl, by
    if op.kind not in (_ffi.OP_CONV, _ffi.OP_WGRAD, _ffi.OP_WGRAD_PARTIAL):
        return 0.0, 0.0
    i = op.i
    B, Hi, Wi, Ho, Wo, K, N, kh, kw, mode = i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7], i[8], i[13]
    pix = Ho * Wo if (mode == 0 or op.kind != _ffi.OP_CONV) else Hi * Wi
    return 2.0 * B * pix * kh * kw * K * N, 4.0 * (B * Hi * Wi * K + B * Ho * Wo * N + kh * kw * K * N)


def tail_stamps(lib, E, mk, feed, args, dev, plain_ms):
    """Where the replayed step spends its end: a STAMPED copy of the plan (Schedule.STAMPS: mh_stamp ops at the start, the end of
    the forward pass, the side lane's first op, the start / end of every filter-gradient batch, the end of the input-gradient chain, the join, the
    end) is captured and replayed args.stamps times; all times in us from the step's first op, median over the replays.  No tracer involved: the
    stamps are kernels of the graph itself (each costs the chain one ~2-5 us launch, `stamped_ms_per_step` tells by how much)."""
    import numpy as np
    e = mk(args.precision, STAMPS=True); feed(e)              # (a Schedule of its own: the stamped plan never leaks into another engine)
    plan = e.build_plan(args.mode, lr=1e-4)
    labels = list(e.stamp_labels)
    rate_khz = float(lib.stamp_rate_khz()) or 1e5
    rows = []
    with dev.ctx():
        plan.run(lib, dev.sh); dev.sync_stream()
        plan.capture(lib, dev.sh)
        for _ in range(5):
            plan.launch(lib, dev.sh)
        dev.sync_stream()
        ms = _time_ms(lib, dev.stream, lambda: plan.launch(lib, dev.sh), 50)
        for _ in range(args.stamps):
            plan.launch(lib, dev.sh); dev.sync_stream()
            rows.append(e.stamps[:len(labels)].cpu().numpy().astype(np.float64))
    t = np.median(np.stack(rows), axis=0)
    us = (t - t[0]) * 1e3 / rate_khz
    at = {lab: float(u) for (lab, _), u in zip(labels, us)}
    out = {"stamps_us": [{"label": lab, "lane": lane, "t_us": float(u)} for (lab, lane), u in zip(labels, us)],
           "stamped_ms_per_step": ms, "plain_ms_per_step": plain_ms, "replays": args.stamps, "clock_khz": rate_khz}
    if "chain_end" in at and "end" in at:
        out["tail_us"] = at["end"] - at["chain_end"]                      # everything behind the last input gradient
        out["join_wait_us"] = at.get("joined", at["chain_end"]) - at["chain_end"]      # of it: lane 0 idle, waiting for the side lane
    if "side_lane_first_op" in at:
        out["side_lane_start_us"] = at["side_lane_first_op"]
    b = sorted((lab, u) for lab, u in at.items() if lab.startswith("wgrad_batch"))
    if b:
        out["first_wgrad_batch_start_us"] = min(u for lab, u in b if lab.endswith("_start"))
        out["last_wgrad_batch_end_us"] = max(u for lab, u in b if lab.endswith("_end"))
    if "forward_end" in at:
        out["forward_us"] = at["forward_end"]
    return out


def corr_rooflines(lib, stream, md=2, C2=32, H=96, W=320, reps=10):
    """SURVEY 8(d) correlation protocol beyond the forward level-2 entry of roofline(): working sets larger than the 256 MiB Infinity Cache, HIP events on
    the launch stream.  Returns
      roofline_corr_bwd       mh_corr_bwd (TF-form gradient, sharedLayers.py:41-51) at the level-2 shape x 64 streams; bytes = B H W (4C + D) 4
      roofline_corr_warp_bwd  mh_corr_warp_bwd -- what the step runs per level: correlation + concat gradient fused with the warp gradient -- same shape,
                              bytes = every operand once: reads g (C + D + 1), L, Rw, the right features (slope taps), u, the accumulate operands dL and
                              dimg; writes dL, dimg, du = B H W (8C + D + 3) 4
      roofline_corr_d81_fwd / _bwd   DispNet's 81-shift volume (md 40, C 128) x 16 streams in the bf16 arithmetic of the 'mixed' mode
                              (mh_corr_fwd_prec / mh_corr_bwd_prec precision 1); bytes = B H W (2C + D) 4 / B H W (4C + D) 4"""
    sh = stream.cuda_stream
    dev = "cuda"
    out = {}

    def ent(kernel, byts, ms, key, extra=None):
        g = byts / (ms * 1e-3) / 1e9
        e = {"kernel": kernel, "bound": "hbm", "achieved": g, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": g / PEAK_HBM_GBS, "traffic": _pmc_traffic(key),
             "traffic_source": _fixed_source(key), "launch_ms": ms, "algorithmic_bytes_per_launch": byts}
        if extra:
            e.update(extra)
        return e

    try:
        B, D = 64, 2 * md + 1
        ld = (C2 + D + 1 + 3) // 4 * 4
        L = torch.randn(B, H, W, C2, device=dev); R = torch.randn(B, H, W, C2, device=dev)
        g = torch.randn(B, H, W, ld, device=dev)
        dL = torch.zeros(B, H, W, C2, device=dev); dR = torch.zeros(B, H, W, C2, device=dev)
        gv = ops.View(g, B, H, W, ld, ld)
        ms = _time_ms(lib, stream, lambda: ops.corr_bwd(lib, gv, ops.view(L), ops.view(R), ops.view(dL), ops.view(dR), md, 1, coff=C2, stream=sh, precision=0), reps)
        out["roofline_corr_bwd"] = ent(lib.last_kernel().decode() + " (B=%d x %dx%dx%d, D=%d)" % (B, H, W, C2, D), float(B) * H * W * (4 * C2 + D) * 4, ms, "roofline_corr_bwd")
        u = (torch.rand(B, H, W, device=dev) - 0.5) * 8.0
        Rw = torch.empty_like(R); du = torch.zeros(B, H, W, device=dev)
        ops.warp_fwd(lib, ops.view(R), u, ops.view(Rw), stream=sh)
        ms = _time_ms(lib, stream, lambda: ops.corr_warp_bwd(lib, gv, ops.view(L), ops.view(Rw), ops.view(R), u, ops.view(dL), ops.view(dR), du, md, 1, coff=C2, acc_l=True,
                                                              copy_left=True, stream=sh), reps)
        kname = lib.last_kernel().decode()
        e = ent(kname + " (B=%d x %dx%dx%d)" % (B, H, W, C2), float(B) * H * W * (8 * C2 + D + 3) * 4, ms, "roofline_corr_warp_bwd")
        # in situ: one stream, cache resident, launch bound -- what the step's level-2 node costs
        L1, R1, Rw1, g1, u1 = (t[:1].contiguous() for t in (L, R, Rw, g, u))
        dL1 = torch.zeros_like(L1); dR1 = torch.zeros_like(L1); du1 = torch.zeros(1, H, W, device=dev)
        gv1 = ops.View(g1, 1, H, W, ld, ld)
        e["in_situ_B1_ms"] = _time_ms(lib, stream, lambda: ops.corr_warp_bwd(lib, gv1, ops.view(L1), ops.view(Rw1), ops.view(R1), u1, ops.view(dL1), ops.view(dR1), du1, md, 1,
                                                                          coff=C2, acc_l=True, copy_left=True, stream=sh), 2 * reps)
        out["roofline_corr_warp_bwd"] = e
        del L, R, g, dL, dR, Rw, u, du
    except Exception as ex:
        out["roofline_corr_bwd"] = {"error": repr(ex)}
    try:
        B, C, mdl = 16, 128, 40
        D = 2 * mdl + 1
        L = torch.randn(B, H, W, C, device=dev); R = torch.randn(B, H, W, C, device=dev)
        vol = torch.empty(B, H, W, D, device=dev)
        ms = _time_ms(lib, stream, lambda: ops.corr_fwd(lib, ops.view(L), ops.view(R), ops.view(vol), mdl, stream=sh, precision=1), reps)
        out["roofline_corr_d81_fwd"] = ent(lib.last_kernel().decode() + " (B=%d x %dx%dx%d, D=%d)" % (B, H, W, C, D), float(B) * H * W * (2 * C + D) * 4, ms, "roofline_corr_d81_fwd")
        ld = (D + 3) // 4 * 4
        g = torch.randn(B, H, W, ld, device=dev)
        dL = torch.empty_like(L); dR = torch.empty_like(R)
        gv = ops.View(g, B, H, W, D, ld)
        ms = _time_ms(lib, stream, lambda: ops.corr_bwd(lib, gv, ops.view(L), ops.view(R), ops.view(dL), ops.view(dR), mdl, 1, coff=0, stream=sh, precision=1), reps)
        e = ent(lib.last_kernel().decode() + " (B=%d x %dx%dx%d, D=%d)" % (B, H, W, C, D), float(B) * H * W * (4 * C + D) * 4, ms, "roofline_corr_d81_bwd")
        ms0 = _time_ms(lib, stream, lambda: ops.corr_bwd(lib, gv, ops.view(L), ops.view(R), ops.view(dL), ops.view(dR), mdl, 1, coff=0, stream=sh, precision=0), max(2, reps // 3))
        e["exact_fp32_launch_ms"] = ms0
        e["exact_fp32_frac"] = float(B) * H * W * (4 * C + D) * 4 / (ms0 * 1e-3) / 1e9 / PEAK_HBM_GBS
        out["roofline_corr_d81_bwd"] = e
    except Exception as ex:
        out["roofline_corr_d81_bwd"] = {"error": repr(ex)}
    return out
