"""Per-layer kernel timing on the MI355X (HIP events), used to tune tile shapes / splits.
usage: python scripts/microbench.py [conv|wgrad|corr|all]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")):
    sys.path.insert(0, p)
import torch
from madnet_hip import _ffi, ops
from madnet_hip.benchtools import _time_ms

lib = _ffi.lib()
stream = torch.cuda.Stream()
what = sys.argv[1] if len(sys.argv) > 1 else "all"
if len(sys.argv) > 2 and sys.argv[2] == "bf16":
    ops.PRECISION = 1
dev = "cuda"

# (name, B, H, W, Cin, Cout, stride, dil)
LAYERS = [("L2 128->128 d2", 1, 96, 320, 128, 128, 1, 2), ("L2 128->96", 1, 96, 320, 128, 96, 1, 1),
          ("L2 96->64", 1, 96, 320, 96, 64, 1, 1), ("L2 64->32", 1, 96, 320, 64, 32, 1, 1), ("L2 38->128", 1, 96, 320, 38, 128, 1, 1),
          ("L3 128->128", 1, 48, 160, 128, 128, 1, 1), ("L4 128->128", 1, 24, 80, 128, 128, 1, 1), ("L5 128->128", 1, 12, 40, 128, 128, 1, 1),
          ("P 16->16 @1/2 x2", 2, 192, 640, 16, 16, 1, 1), ("P 16->32 s2 x2", 2, 192, 640, 16, 32, 2, 1), ("P 32->32 @1/4 x2", 2, 96, 320, 32, 32, 1, 1),
          ("P 64->64 @1/8 x2", 2, 48, 160, 64, 64, 1, 1)]
TILES = [(0, 0, 0), (128, 64, 0), (64, 64, 0), (32, 64, 64), (32, 64, 128), (64, 32, 64), (64, 32, 128), (32, 32, 64), (32, 32, 128)]
if ops.PRECISION == 1:      # bf16 instances all have KT = 64
    TILES = [(0, 0, 0), (128, 128, 64), (128, 64, 64), (64, 128, 64), (64, 64, 64), (128, 32, 64), (32, 128, 64), (32, 64, 64), (64, 32, 64), (32, 32, 64)]


def run_conv():
    print("%-20s %-6s %s" % ("layer", "mode", " ".join("%10s" % ("auto" if t[0] == 0 else "%dx%d/%d" % t) for t in TILES)) + "   GFLOP")
    for name, B, H, W, Ci, Co, s, d in LAYERS:
        ld = (Ci + 3) // 4 * 4
        x = torch.randn(B, H, W, ld, device=dev); xv = ops.View(x, B, H, W, Ci, ld)
        w = torch.randn(3, 3, Ci, Co, device=dev) * 0.05; b = torch.randn(Co, device=dev)
        Ho, Wo, _, _ = ops.conv_geometry(H, W, 3, 3, s, d)
        y = torch.empty(B, Ho, Wo, Co, device=dev); dx = torch.empty(B, H, W, ld, device=dev)
        flops = 2.0 * B * Ho * Wo * 9 * Ci * Co
        for mode in ("fwd", "dgrad"):
            res = []
            for bm, bn, kt in TILES:
                if bn and ((mode == "fwd" and bn > max(16, Co) * 2) or (mode == "dgrad" and bn > max(16, Ci) * 2)):
                    res.append(None); continue
                lib.tune_conv_tile(bm, bn | (kt << 16))
                try:
                    with torch.cuda.stream(stream):
                        if mode == "fwd":
                            fn = lambda: ops.conv2d_fwd(lib, xv, w, b, ops.view(y), stride=s, dil=d, alpha=0.2, stream=stream.cuda_stream)
                        else:
                            fn = lambda: ops.conv2d_dgrad(lib, ops.view(y), w, ops.View(dx, B, H, W, Ci, ld), stride=s, dil=d, accumulate=True,
                                                          mask_ref=xv, mask_alpha=0.2, stream=stream.cuda_stream)
                        res.append(_time_ms(lib, stream, fn, 10) * 1e3)
                except Exception as e:
                    res.append(None)
            lib.tune_conv_tile(0, 0)
            print("%-20s %-6s %s   %.2f  (best %.0f TF/s)" % (name, mode, " ".join("%10s" % ("-" if r is None else "%.1f" % r) for r in res), flops / 1e9,
                                                            flops / (min(r for r in res if r) * 1e-6) / 1e12))


def run_wgrad():
    targets = [192, 384, 512, 768, 1536]
    print("%-20s %s" % ("wgrad layer", " ".join("%9d" % t for t in targets)))
    for name, B, H, W, Ci, Co, s, d in LAYERS:
        ld = (Ci + 3) // 4 * 4
        x = torch.randn(B, H, W, ld, device=dev); xv = ops.View(x, B, H, W, Ci, ld)
        Ho, Wo, _, _ = ops.conv_geometry(H, W, 3, 3, s, d)
        dz = torch.randn(B, Ho, Wo, Co, device=dev)
        dw = torch.zeros(3, 3, Ci, Co, device=dev); db = torch.zeros(Co, device=dev)
        flops = 2.0 * B * Ho * Wo * 9 * Ci * Co
        res = []
        for t in targets:
            lib.tune_wgrad_wgs(t)
            with torch.cuda.stream(stream):
                res.append(_time_ms(lib, stream, lambda: ops.conv2d_wgrad(lib, xv, ops.view(dz), dw, db, stride=s, dil=d, stream=stream.cuda_stream), 10) * 1e3)
        lib.tune_wgrad_wgs(0)
        print("%-20s %s   (best %.0f TF/s)" % (name, " ".join("%9.1f" % r for r in res), flops / (min(res) * 1e-6) / 1e12))


def run_wgradp():
    """atomic-free filter gradient as the engines run it (bf16 operands): partial launch into the split workspace + its share of the
    reduction, for the default 128x128 tiles and the 64x64-tile experiment (4x fewer pixel splits for the same workgroup count)."""
    PL = [("L2 128->128", 1, 96, 320, 128, 128, 1), ("L2 128->128 d4", 1, 96, 320, 128, 128, 4), ("L2 128->96", 1, 96, 320, 128, 96, 1),
          ("L2 96->64", 1, 96, 320, 96, 64, 1), ("L2 38->128", 1, 96, 320, 38, 128, 1), ("L3 128->128", 1, 48, 160, 128, 128, 1)]
    ops.PRECISION = 1
    print("%-16s %s" % ("wgrad layer", "  [tile: partial us + reduce us = total, splits, workspace MB]"))
    for name, B, H, W, Ci, Co, d in PL:
        ld = (Ci + 3) // 4 * 4
        x = torch.randn(B, H, W, ld, device=dev); xv = ops.View(x, B, H, W, Ci, ld)
        dz = torch.randn(B, H, W, Co, device=dev)
        out = []
        for tile, tgt in (("128", 0), ("128/768wg", 768), ("64", 100000), ("128 8w 2x4", 200000), ("128 8w 4x2", 300000), ("128 8w 2x4/768", 200768)):
            lib.tune_wgrad_wgs(tgt)
            dw = torch.zeros(3, 3, Ci, Co, device=dev); db = torch.zeros(Co, device=dev)
            wsa = ops.WgradWorkspace(dev); segs, keep = [], []
            ops.conv2d_wgrad_partial(lib, lib, wsa, segs, xv, ops.view(dz), dw, db, dil=d)
            if not segs:
                out.append("%s: single split" % tile); continue
            ws, dst, size, splits = segs[0]
            desc = ops.conv_desc(B, H, W, H, W, Ci, Co, 3, 3, 1, d, d, d, 0, 0, ld, Co, precision=1)
            sp = C.c_int32(splits)
            with torch.cuda.stream(stream):
                tp = _time_ms(lib, stream, lambda: lib.conv2d_wgrad_partial(C.byref(desc), C.c_void_p(x.data_ptr()), C.c_void_p(dz.data_ptr()), Co, C.c_void_p(ws),
                                                                            C.byref(sp), C.c_void_p(db.data_ptr()), C.c_void_p(stream.cuda_stream)), 20) * 1e3
            arr = (_ffi.WgradSeg * 1)(); arr[0].ws, arr[0].dst, arr[0].size, arr[0].splits, arr[0].blk0, arr[0].accumulate = ws, dst, size, splits, 0, 0
            table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            with torch.cuda.stream(stream):
                tr = _time_ms(lib, stream, lambda: lib.wgrad_reduce(C.c_void_p(table.data_ptr()), 1, (size + 1023) // 1024, C.c_void_p(stream.cuda_stream)), 20) * 1e3
            out.append("%s: %.1f + %.1f = %.1f, %d, %.1f" % (tile, tp, tr, tp + tr, splits, splits * size * 4 / 1e6))
        lib.tune_wgrad_wgs(0)
        print("%-16s %s" % (name, "  |  ".join(out)))
    ops.PRECISION = 0


def run_corr():
    for (B, H, W, Cc, md) in [(64, 96, 320, 32, 2), (16, 96, 320, 32, 2), (1, 96, 320, 32, 2), (64, 48, 160, 64, 2), (16, 96, 320, 128, 40)]:
        L = torch.randn(B, H, W, Cc, device=dev); R = torch.randn(B, H, W, Cc, device=dev)
        D = 2 * md + 1
        out = torch.empty(B, H, W, D, device=dev)
        byts = float(B) * H * W * (2 * Cc + D) * 4
        for direct in (0, 1):
            lib.tune_corr(direct)
            with torch.cuda.stream(stream):
                ms = _time_ms(lib, stream, lambda: ops.corr_fwd(lib, ops.view(L), ops.view(R), ops.view(out), md, stream=stream.cuda_stream), 10)
            print("corr fwd %s B=%d %dx%dx%d D=%d: %.1f us  %.0f GB/s (%.1f%% of 8 TB/s)" % ("direct" if direct else "lds   ", B, H, W, Cc, D, ms * 1e3, byts / ms / 1e6, byts / ms / 1e6 / 80))
        lib.tune_corr(1)
    # the large-shift (DispNet) volume by arithmetic mode: exact fp32 MFMA / bf16 MFMA / split-bf16 (mh_corr_fwd_prec)
    for (B, H, W, Cc, md) in [(16, 96, 320, 128, 40), (1, 96, 320, 128, 40)]:
        L = torch.randn(B, H, W, Cc, device=dev); R = torch.randn(B, H, W, Cc, device=dev)
        D = 2 * md + 1
        out = torch.empty(B, H, W, D, device=dev)
        byts = float(B) * H * W * (2 * Cc + D) * 4
        for prec, nm in ((0, "fp32 "), (1, "bf16 "), (2, "bf16x3")):
            with torch.cuda.stream(stream):
                ms = _time_ms(lib, stream, lambda: ops.corr_fwd(lib, ops.view(L), ops.view(R), ops.view(out), md, stream=stream.cuda_stream, precision=prec), 10)
            print("corr fwd %s B=%d %dx%dx%d D=%d: %.1f us  %.0f GB/s (%.1f%% of 8 TB/s)  [%s]" % (nm, B, H, W, Cc, D, ms * 1e3, byts / ms / 1e6, byts / ms / 1e6 / 80, lib.last_kernel().decode()))
    # gradient of the large-shift (DispNet) volume: gather kernel (tune_corr 0) vs the banded-GEMM MFMA pair (default)
    for (B, H, W, Cc, md) in [(1, 96, 320, 128, 40), (16, 96, 320, 128, 40)]:
        L = torch.randn(B, H, W, Cc, device=dev); R = torch.randn(B, H, W, Cc, device=dev)
        D = 2 * md + 1; ld = (D + 3) // 4 * 4
        g = torch.randn(B, H, W, ld, device=dev); dL = torch.empty_like(L); dR = torch.empty_like(R)
        byts = float(B) * H * W * (4 * Cc + D) * 4
        for direct in (0, 1):
            lib.tune_corr(direct)
            with torch.cuda.stream(stream):
                ms = _time_ms(lib, stream, lambda: ops.corr_bwd(lib, ops.View(g, B, H, W, D, ld), ops.view(L), ops.view(R), ops.view(dL), ops.view(dR), md, 1,
                                                                stream=stream.cuda_stream), 10)
            print("corr bwd %s B=%d %dx%dx%d D=%d: %.1f us  %.0f GB/s (%.1f%% of 8 TB/s)" % ("mfma  " if direct else "gather", B, H, W, Cc, D, ms * 1e3, byts / ms / 1e6, byts / ms / 1e6 / 80))
        lib.tune_corr(1)


def run_patch():
    """bf16 stride-1 3x3 layers: gather kernel (mode 0) vs the patch-staged kernel (128-px tile, its 8-wave variant, 64-px tile)."""
    ops.PRECISION = 1
    MODES = [0, 128, 128 + 256, 64]
    PL = [("L2 128->128 d1", 1, 96, 320, 128, 128, 1), ("L2 128->128 d2", 1, 96, 320, 128, 128, 2), ("L2 128->128 d4", 1, 96, 320, 128, 128, 4),
          ("L2 128->128 d8", 1, 96, 320, 128, 128, 8), ("L2 128->128 d16", 1, 96, 320, 128, 128, 16), ("L2 128->96", 1, 96, 320, 128, 96, 1),
          ("L2 96->64", 1, 96, 320, 96, 64, 1), ("L2 40->128", 1, 96, 320, 40, 128, 1), ("L3 128->128", 1, 48, 160, 128, 128, 1),
          ("L3 128->96", 1, 48, 160, 128, 96, 1), ("L4 128->128", 1, 24, 80, 128, 128, 1), ("L2 128->128 B4", 4, 96, 320, 128, 128, 1)]
    print("%-18s %-6s %s" % ("layer (bf16)", "mode", " ".join("%10s" % ("gather" if m == 0 else "patch%d%s" % (m & 255, "w8" if m & 256 else "")) for m in MODES)))
    for name, B, H, W, Ci, Co, d in PL:
        x = torch.randn(B, H, W, Ci, device=dev); xv = ops.view(x)
        w = torch.randn(3, 3, Ci, Co, device=dev) * 0.05; b = torch.randn(Co, device=dev)
        y = torch.empty(B, H, W, Co, device=dev); dx = torch.empty(B, H, W, Ci, device=dev)
        flops = 2.0 * B * H * W * 9 * Ci * Co
        for mode in ("fwd", "dgrad"):
            res = []
            for m in MODES:
                lib.tune_conv_patch(m)
                with torch.cuda.stream(stream):
                    if mode == "fwd":
                        fn = lambda: ops.conv2d_fwd(lib, xv, w, b, ops.view(y), stride=1, dil=d, alpha=0.2, stream=stream.cuda_stream)
                    else:
                        fn = lambda: ops.conv2d_dgrad(lib, ops.view(y), w, ops.view(dx), stride=1, dil=d, accumulate=True, mask_ref=xv, mask_alpha=0.2,
                                                      stream=stream.cuda_stream)
                    res.append(_time_ms(lib, stream, fn, 10) * 1e3)
            lib.tune_conv_patch(-1)
            print("%-18s %-6s %s   (best %.0f TF/s)" % (name, mode, " ".join("%10.1f" % r for r in res), flops / (min(res) * 1e-6) / 1e12))
    ops.PRECISION = 0


def run_patchdbg():
    """Where the time of the patch kernel goes: full kernel vs no K walk vs no patch staging vs neither (launch + epilogue)."""
    ops.PRECISION = 1
    for base in (128 + 256, 64):
        MODES = [base, base + 512, base + 1024, base + 1536]
        print("tile mode %d: %-14s %s" % (base, "layer", " ".join("%10s" % n for n in ("full", "noK", "nostage", "neither"))))
        for name, B, H, W, Ci, Co, d in [("L2 128->128", 1, 96, 320, 128, 128, 1), ("L4 128->128", 1, 24, 80, 128, 128, 1), ("1 tile", 1, 8, 16, 128, 128, 1)]:
            x = torch.randn(B, H, W, Ci, device=dev); xv = ops.view(x)
            w = torch.randn(3, 3, Ci, Co, device=dev) * 0.05; b = torch.randn(Co, device=dev)
            y = torch.empty(B, H, W, Co, device=dev)
            res = []
            for m in MODES:
                lib.tune_conv_patch(m)
                with torch.cuda.stream(stream):
                    res.append(_time_ms(lib, stream, lambda: ops.conv2d_fwd(lib, xv, w, b, ops.view(y), stride=1, dil=d, alpha=0.2, stream=stream.cuda_stream), 20) * 1e3)
            lib.tune_conv_patch(-1)
            print("              %-14s %s" % (name, " ".join("%10.1f" % r for r in res)))
    # launch floor: an (almost) empty kernel through the same path
    t = torch.zeros(64, device=dev)
    with torch.cuda.stream(stream):
        print("launch floor (fill 64 floats): %.1f us" % (_time_ms(lib, stream, lambda: lib.fill(t.data_ptr(), 64, 0.0, stream.cuda_stream), 20) * 1e3))
    ops.PRECISION = 0


def run_x3dbg():
    """split-bf16 (precision code 2) patch kernel: phase breakdown (full / no K walk / no staging / neither) and the generic-K
    instance, next to plain bf16 and exact fp32 on the same layers."""
    PL = [("L2 128->128 d1", 1, 96, 320, 128, 128, 1), ("L2 128->128 d2", 1, 96, 320, 128, 128, 2), ("L2 128->96", 1, 96, 320, 128, 96, 1),
          ("L2 96->64", 1, 96, 320, 96, 64, 1), ("L2 40->128", 1, 96, 320, 40, 128, 1), ("L3 128->128", 1, 48, 160, 128, 128, 1),
          ("L3 72->128", 1, 48, 160, 72, 128, 1), ("L4 128->128", 1, 24, 80, 128, 128, 1)]
    print("%-16s %9s %9s %9s %9s %9s | %9s %9s" % ("layer (fwd)", "x3 full", "x3 noK", "x3 nostg", "x3 none", "x3 genK", "bf16", "fp32"))
    for name, B, H, W, Ci, Co, d in PL:
        x = torch.randn(B, H, W, Ci, device=dev); xv = ops.view(x)
        w = torch.randn(3, 3, Ci, Co, device=dev) * 0.05; b = torch.randn(Co, device=dev)
        y = torch.empty(B, H, W, Co, device=dev)
        flops = 2.0 * B * H * W * 9 * Ci * Co
        res = []
        for code, m in ((2, 128), (2, 128 + 512), (2, 128 + 1024), (2, 128 + 1536), (2, 128 + 2048), (1, -1), (0, -1)):
            lib.tune_conv_patch(m)
            ops.PRECISION = code
            with torch.cuda.stream(stream):
                res.append(_time_ms(lib, stream, lambda: ops.conv2d_fwd(lib, xv, w, b, ops.view(y), stride=1, dil=d, alpha=0.2, stream=stream.cuda_stream), 20) * 1e3)
            ops.PRECISION = 0
        lib.tune_conv_patch(-1)
        print("%-16s %s | %9.1f %9.1f   (x3 %.0f TF/s algorithmic)" % (name, " ".join("%9.1f" % r for r in res[:5]), res[5], res[6], flops / (res[0] * 1e-6) / 1e12))


def run_bank():
    """split-bf16 forward: LDS-staged weights (conv_patch_kernel X3) vs the fragment-bank kernel (mh_conv2d_wb), with the bank kernel's
    phase breakdown (no K walk / no staging) and the pack launch for the 13 bank layers of MADNet."""
    PL = [("L2 128->128 d1", 1, 96, 320, 128, 128, 1), ("L2 128->128 d2", 1, 96, 320, 128, 128, 2), ("L2 128->128 d8", 1, 96, 320, 128, 128, 8),
          ("L2 128->96", 1, 96, 320, 128, 96, 1), ("L2 96->64", 1, 96, 320, 96, 64, 1), ("L2 38->128", 1, 96, 320, 38, 128, 1),
          ("L3 128->128", 1, 48, 160, 128, 128, 1), ("L3 70->128", 1, 48, 160, 70, 128, 1), ("L4 128->128", 1, 24, 80, 128, 128, 1)]
    print("%-16s %9s | %9s %9s %9s %9s %9s   %s" % ("layer (fwd)", "x3 LDS", "x3 bank", "bank noK", "bank nostg", "bank 128px", "bank 4w", "max |diff|"))
    for name, B, H, W, Ci, Co, d in PL:
        ld = (Ci + 3) // 4 * 4
        xb = torch.zeros(B, H, W, ld, device=dev); xb[..., :Ci] = torch.randn(B, H, W, Ci, device=dev); xv = ops.View(xb, B, H, W, Ci, ld)
        w = torch.randn(3, 3, Ci, Co, device=dev) * 0.05; b = torch.randn(Co, device=dev)
        y = torch.empty(B, H, W, Co, device=dev); y2 = torch.empty(B, H, W, Co, device=dev)
        bank = torch.zeros(ops.pack_bytes(w) // 4, device=dev); keep = []
        ops.pack_weights(lib, [(w, bank)], dev, keep)
        lib.tune_conv_bank(0)         # the 64x128 / 128x64 bank kernel on every row (L4 would otherwise go to the small-layer kernel)
        flops = 2.0 * B * H * W * 9 * Ci * Co
        res = []
        ops.PRECISION = 2
        for m, wb, out in ((128, None, y), (128, bank, y2), (128 + 512, bank, y2), (128 + 1024, bank, y2), (128 + 0x8000, bank, y2), (128 + 0x10000, bank, y2)):
            lib.tune_conv_patch(m)
            with torch.cuda.stream(stream):
                res.append(_time_ms(lib, stream, lambda: ops.conv2d_fwd(lib, xv, w, b, ops.view(out), stride=1, dil=d, alpha=0.2, stream=stream.cuda_stream, wb=wb), 20) * 1e3)
        lib.tune_conv_patch(128)
        ops.conv2d_fwd(lib, xv, w, b, ops.view(y2), stride=1, dil=d, alpha=0.2, wb=bank)
        ops.conv2d_fwd(lib, xv, w, b, ops.view(y), stride=1, dil=d, alpha=0.2)
        torch.cuda.synchronize()
        ops.PRECISION = 0
        lib.tune_conv_patch(-1)
        lib.tune_conv_bank(-1)
        print("%-16s %9.1f | %9.1f %9.1f %9.1f %9.1f %9.1f   %.3g   (bank %.0f TF/s algorithmic, %s)" % (name, res[0], res[1], res[2], res[3], res[4], res[5], (y - y2).abs().max().item(),
              flops / (res[1] * 1e-6) / 1e12, lib.last_kernel().decode()[:60]))
    shapes = [(38, 128), (128, 128), (128, 96), (96, 64), (70, 128), (128, 128), (128, 96), (96, 64), (33, 128), (128, 128), (128, 128), (128, 96), (96, 64)]
    ws = [torch.randn(3, 3, k, n, device=dev) for k, n in shapes]
    banks = [torch.zeros(ops.pack_bytes(w) // 4, device=dev) for w in ws]
    keep = []
    with torch.cuda.stream(stream):
        t = _time_ms(lib, stream, lambda: ops.pack_weights(lib, list(zip(ws, banks)), dev, keep, stream=stream.cuda_stream), 20) * 1e3
    print("mh_pack_weights, 13 layers (%.1f MB of banks): %.1f us" % (sum(b.numel() * 4 for b in banks) / 1e6, t))


def run_planes():
    """split-bf16 forward from pre-split planes (mh_conv2d_planes, csrc/conv_planes.hip) against the fragment-bank kernel on the same layer:
    tile variants, phase breakdown (no K walk / no staging), planes-only vs fp32 + planes output, the split launch."""
    PL = [("L2 128->128 d1", 1, 96, 320, 128, 128, 1), ("L2 128->128 d2", 1, 96, 320, 128, 128, 2), ("L2 128->128 d4", 1, 96, 320, 128, 128, 4),
          ("L2 128->96 d8", 1, 96, 320, 128, 96, 8), ("L2 96->64 d16", 1, 96, 320, 96, 64, 16),
          ("L2 128->96", 1, 96, 320, 128, 96, 1), ("L2 96->64", 1, 96, 320, 96, 64, 1), ("L2 64->32", 1, 96, 320, 64, 32, 1), ("L2 38->128", 1, 96, 320, 38, 128, 1),
          ("L2 33->128", 1, 96, 320, 33, 128, 1), ("P 32->32 x2", 2, 96, 320, 32, 32, 1),
          ("L3 128->128", 1, 48, 160, 128, 128, 1), ("L3 70->128", 1, 48, 160, 70, 128, 1), ("L3 96->64", 1, 48, 160, 96, 64, 1), ("P 64->64 x2", 2, 48, 160, 64, 64, 1)]
    only = os.environ.get("MB_ONLY")
    print("%-16s %8s | %8s | %8s %8s %8s %8s %8s %8s | %8s %8s   %s" % ("layer (fwd)", "bank", "auto", "32c 128p", "32c 64p", "16c 128p", "16c 64p", "32c 32p", "16c 32p", "auto noK", "auto pl-only", "max|diff| vs bank"))
    for name, B, H, W, Ci, Co, d in PL:
        if only and only not in name:
            continue
        ld = (Ci + 3) // 4 * 4
        xb = torch.zeros(B, H, W, ld, device=dev); xb[..., :Ci] = torch.randn(B, H, W, Ci, device=dev); xv = ops.View(xb, B, H, W, Ci, ld)
        w = torch.randn(3, 3, Ci, Co, device=dev) * 0.05; b = torch.randn(Co, device=dev)
        y = torch.empty(B, H, W, Co, device=dev); y2 = torch.empty(B, H, W, Co, device=dev)
        keep = []
        bank = torch.zeros(ops.pack_bytes(w) // 4, device=dev)
        bank32 = torch.zeros(ops.pack_bytes(w, 2, 2) // 4, device=dev)
        ops.pack_weights(lib, [(w, bank), (w, bank32, 2, 2)], dev, keep)
        xp = ops.Planes(ops.Shadow(B, H, W, Ci, dev), dev)
        yp = ops.Planes(ops.Shadow(B, H, W, Co, dev), dev)
        ops.plane_split(lib, [(xv, xp)], dev, keep)
        sh = stream.cuda_stream
        lib.tune_conv_bank(0); lib.tune_conv_patch(128)
        ops.PRECISION = 2
        with torch.cuda.stream(stream):
            t_bank = _time_ms(lib, stream, lambda: ops.conv2d_fwd(lib, xv, w, b, ops.view(y), stride=1, dil=d, alpha=0.2, stream=sh, wb=bank), 20) * 1e3
        ops.PRECISION = 0
        lib.tune_conv_patch(-1); lib.tune_conv_bank(-1)
        res = []
        for mode, outv, outp in ((0, y2, yp), (1, y2, yp), (2, y2, yp), (3, y2, yp), (4, y2, yp), (5, y2, yp), (6, y2, yp), (256, y2, yp), (0, None, yp)):
            lib.tune_conv_planes(mode)
            try:
                with torch.cuda.stream(stream):
                    res.append(_time_ms(lib, stream, lambda: ops.conv2d_planes(lib, xp, w, bank32, b, out=(ops.view(outv) if outv is not None else None), out_planes=outp,
                                                                               dil=d, alpha=0.2, stream=sh), 20) * 1e3)
            except Exception as e:
                res.append(float("nan"))
        lib.tune_conv_planes(0)
        ops.conv2d_planes(lib, xp, w, bank32, b, out=ops.view(y2), out_planes=yp, dil=d, alpha=0.2)
        kn = lib.last_kernel().decode()
        torch.cuda.synchronize()
        flops = 2.0 * B * H * W * 9 * Ci * Co
        best = min(r for r in res[1:7] if r == r)
        print("%-16s %8.1f | %8.1f | %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f | %8.1f %8.1f   %.3g  (auto %.0f TF/s algorithmic = %.3f of 2.5 PF, best forced %.1f; %s)"
              % (name, t_bank, res[0], res[1], res[2], res[3], res[4], res[5], res[6], res[7], res[8], (y - y2).abs().max().item(), flops / (res[0] * 1e-6) / 1e12,
                 flops / (res[0] * 1e-6) / 2.5e15, best, kn[:64]))


def run_dispnet():
    """DispNet's stride-1 3x3 layers at 384x1280 (Nets/DispNet.py:75-152): the igemm kernels the engine ran in round 3 (bf16 / split-bf16 forward, bf16 input
    gradient) against mh_conv2d_planes / mh_conv2d_planes_bwd (K-chunked beyond 128 channels)."""
    DL = [("conv3_1 256->256", 48, 160, 256, 256, 1), ("conv4_1 512->512", 24, 80, 512, 512, 1), ("conv5_1 512->512", 12, 40, 512, 512, 1),
          ("conv6_1 1024->1024", 6, 20, 1024, 1024, 1), ("iconv5 1025->512", 12, 40, 1025, 512, 1), ("iconv4 769->256", 24, 80, 769, 256, 1),
          ("iconv3 385->128", 48, 160, 385, 128, 1), ("iconv2 193->64 x3", 96, 320, 193, 64, 2), ("iconv1 97->32 x3", 192, 640, 97, 32, 2)]
    only = os.environ.get("MB_ONLY")
    print("%-22s | %9s %9s %7s | %9s %9s %7s |  %s" % ("layer", "fwd igemm", "fwd planes", "TF/s", "dgrad ig.", "dg planes", "TF/s", "kernels"))
    sh = stream.cuda_stream
    for name, H, W, Ci, Co, code in DL:
        if only and only not in name:
            continue
        B = 1
        ld = (Ci + 7) // 8 * 8
        xb = torch.zeros(B, H, W, ld, device=dev); xb[..., :Ci] = torch.randn(B, H, W, Ci, device=dev); xv = ops.View(xb, B, H, W, Ci, ld)
        w = torch.randn(3, 3, Ci, Co, device=dev) * 0.03; b = torch.randn(Co, device=dev)
        y = torch.empty(B, H, W, Co, device=dev); y2 = torch.empty(B, H, W, Co, device=dev)
        dz = torch.randn(B, H, W, Co, device=dev); dx = torch.zeros(B, H, W, ld, device=dev); dx2 = torch.zeros(B, H, W, ld, device=dev)
        dxv, dx2v = ops.View(dx, B, H, W, Ci, ld), ops.View(dx2, B, H, W, Ci, ld)
        keep = []
        pl = 2 if code == 2 else 1
        bankf = torch.zeros(ops.pack_bytes(w, pl, 2) // 4, device=dev); bankb = torch.zeros(ops.pack_bytes(w, 1, 3) // 4, device=dev)
        ops.pack_weights(lib, [(w, bankf, pl, 2), (w, bankb, 1, 3)], dev, keep)
        xp = ops.Planes(ops.Shadow(B, H, W, Ci, dev), dev); yp = ops.Shadow(B, H, W, Co, dev)
        ops.plane_split(lib, [(xv, xp)], dev, keep)
        dzs = ops.Shadow(B, H, W, Co, dev); ops.shadow_cast(lib, [(ops.view(dz), dzs)], dev, keep)
        flops = 2.0 * B * H * W * 9 * Ci * Co
        with torch.cuda.stream(stream):
            ops.PRECISION = code
            t0 = _time_ms(lib, stream, lambda: ops.conv2d_fwd(lib, xv, w, b, ops.view(y), alpha=0.1, stream=sh), 10) * 1e3
            k0 = lib.last_kernel().decode()
            ops.PRECISION = 0
            t1 = _time_ms(lib, stream, lambda: ops.conv2d_planes(lib, xp, w, bankf, b, out=ops.view(y2), out_planes=yp, alpha=0.1, stream=sh, bf16=(pl == 1)), 10) * 1e3
            k1 = lib.last_kernel().decode()
            ops.PRECISION_BWD = 1; ops.PRECISION = 1
            t2 = _time_ms(lib, stream, lambda: ops.conv2d_dgrad(lib, ops.view(dz), w, dxv, mask_ref=xv, mask_alpha=0.1, stream=sh), 10) * 1e3
            ops.PRECISION_BWD = None; ops.PRECISION = 0
            t3 = _time_ms(lib, stream, lambda: ops.conv2d_planes_bwd(lib, dzs, w, bankb, dx=dx2v, mask_shadow=xp.hi, mask_alpha=0.1, stream=sh), 10) * 1e3
            k3 = lib.last_kernel().decode()
        torch.cuda.synchronize()
        e_f = (y - y2).abs().max().item() / max(1.0, y.abs().max().item()); e_b = (dx - dx2).abs().max().item() / max(1.0, dx.abs().max().item())
        print("%-22s | %9.1f %9.1f %7.0f | %9.1f %9.1f %7.0f |  fwd %s | bwd %s | rel diff fwd %.2g bwd %.2g | was %s"
              % (name, t0, t1, flops / (t1 * 1e-6) / 1e12, t2, t3, flops / (t3 * 1e-6) / 1e12, k1[22:100], k3[22:100], e_f, e_b, k0[:60]))


def run_planes_phases():
    """where a plane-kernel launch spends its time: the biggest MADNet layer (128 -> 128 at 96x320) with phases removed (mh_tune_conv_planes bits
    8 = no K walk, 9 = no staging, 12 = no epilogue, 13 = epilogue without its stores), fp32 + planes output and planes only"""
    B, H, W, Ci, Co = 1, 96, 320, 128, 128
    x = torch.randn(B, H, W, Ci, device=dev); w = torch.randn(3, 3, Ci, Co, device=dev) * 0.05; b = torch.randn(Co, device=dev)
    keep = []
    bank32 = torch.zeros(ops.pack_bytes(w, 2, 2) // 4, device=dev)
    ops.pack_weights(lib, [(w, bank32, 2, 2)], dev, keep)
    xp = ops.Planes(ops.Shadow(B, H, W, Ci, dev), dev); yp = ops.Planes(ops.Shadow(B, H, W, Co, dev), dev)
    ops.plane_split(lib, [(ops.view(x), xp)], dev, keep)
    y = torch.empty(B, H, W, Co, device=dev)
    sh = stream.cuda_stream
    for label, bits in (("full", 0), ("no K walk", 1), ("no staging", 2), ("no epilogue", 16), ("no stores", 32), ("no K walk, no epilogue", 17), ("no K walk, no stores", 33),
                        ("no K walk, no staging, no epilogue (launch + weight ring)", 19), ("no K walk, no staging", 3)):
        res = []
        for outv in (y, None):
            lib.tune_conv_planes(bits << 8)
            with torch.cuda.stream(stream):
                res.append(_time_ms(lib, stream, lambda: ops.conv2d_planes(lib, xp, w, bank32, b, out=(ops.view(outv) if outv is not None else None), out_planes=yp,
                                                                           alpha=0.2, stream=sh), 20) * 1e3)
        lib.tune_conv_planes(0)
        print("%-62s fp32 + planes %6.1f us    planes only %6.1f us" % (label, res[0], res[1]))


if what == "phases":
    run_planes_phases()
if what == "dispnet":
    run_dispnet()
if what == "planes":
    run_planes()
if what == "wgradp":
    run_wgradp()
if what == "bank":
    run_bank()
if what == "x3dbg":
    run_x3dbg()
if what == "patchdbg":
    run_patchdbg()
if what == "patch":
    run_patch()
if what in ("conv", "all"):
    run_conv()
if what in ("wgrad", "all"):
    run_wgrad()
if what in ("corr", "all"):
    run_corr()
