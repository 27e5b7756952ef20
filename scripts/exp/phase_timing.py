"""Where do the 9-17 us of a 32x32-tile implicit-GEMM launch go?  Runs the small layers of the 1/16-1/64 levels on the -DMH_PHASE_TIMING build
(scripts/exp/build_phase_timing.sh) and prints, per layer, the s_memtime phase stamps of its workgroups next to the launch's HIP-event time.
Phases: entry -> tap tables + geometry (t1) -> first K-tile in LDS (t2) -> K loop done (t3) -> partial tiles staged (t4) -> stores acknowledged (t5)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")):
    sys.path.insert(0, p)
import torch
from madnet_hip import _ffi, ops
from madnet_hip.benchtools import _time_ms

lib = _ffi.Lib(os.path.join(ROOT, "scripts", "exp", "libmadnet_hip_phase.so"))   # built by scripts/exp/build_phase_timing.sh
lib.ensure_init()
dll = lib.dll if hasattr(lib, "dll") else lib._dll
dll.mh_tune_conv_dbg.argtypes = [C.c_void_p]; dll.mh_tune_conv_dbg.restype = C.c_int
dev = "cuda"
stream = torch.cuda.Stream()
LAYERS = [("L6 197->128", 6, 20, 197, 128, 0), ("L6 128->128", 6, 20, 128, 128, 0), ("L6 96->64", 6, 20, 96, 64, 0), ("L5 128->128", 12, 40, 128, 128, 0),
          ("L4 128->128", 24, 80, 128, 128, 0), ("L4 128->128 dgrad", 24, 80, 128, 128, 1), ("L3 128->128 dgrad", 48, 160, 128, 128, 1)]
for prec in (1, 0):
    print("precision code %d" % prec)
    for name, H, W, Ci, Co, dg in LAYERS:
        ld = (Ci + 3) // 4 * 4
        xb = torch.zeros(1, H, W, ld, device=dev); xb[..., :Ci] = torch.randn(1, H, W, Ci, device=dev); xv = ops.View(xb, 1, H, W, Ci, ld)
        w = torch.randn(3, 3, Ci, Co, device=dev) * 0.05; b = torch.randn(Co, device=dev)
        y = torch.zeros(1, H, W, Co, device=dev)
        dx = torch.zeros(1, H, W, ld, device=dev); dxv = ops.View(dx, 1, H, W, Ci, ld)
        buf = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
        ops.PRECISION = prec; ops.PRECISION_BWD = prec
        if dg:
            fn = lambda: ops.conv2d_dgrad(lib, ops.view(y), w, dxv, stream=stream.cuda_stream)
        else:
            fn = lambda: ops.conv2d_fwd(lib, xv, w, b, ops.view(y), alpha=0.2, stream=stream.cuda_stream)
        # cold-ish: flush L2 between runs by touching a big buffer
        big = torch.empty(64 << 20, device=dev)
        with torch.cuda.stream(stream):
            t_ev = _time_ms(lib, stream, fn, 20) * 1e3
            res = []
            for rep in range(10):
                if rep >= 5:
                    big.fill_(1.0)          # reps 5..9: L2 / Infinity Cache flushed (cold); reps 0..4: warm
                buf.zero_()
                stream.synchronize()
                dll.mh_tune_conv_dbg(C.c_void_p(buf.data_ptr()))
                fn()
                stream.synchronize()
                dll.mh_tune_conv_dbg(None)
                res.append(buf.view(-1, 8).cpu().clone())
        ops.PRECISION = 0; ops.PRECISION_BWD = None
        line = "%-20s %6.1f us (events, warm) %s" % (name, t_ev, lib.last_kernel().decode()[:44])
        for tag, r in (("warm", res[4]), ("cold", res[9])):
            r = r[r[:, 0] != 0].double()
            if r.numel() == 0:
                line += " | %s: no stamps" % tag; continue
            tick = (r[:, 7] - r[:, 6]).clamp(min=1) * 10.0 / (r[:, 5] - r[:, 0]).clamp(min=1)     # ns per s_memtime tick (s_memrealtime = 100 MHz)
            ns = tick.median().item()
            t0 = r[:, 0].min()
            ph = [(r[:, i] - r[:, i - 1]).mean().item() * ns / 1e3 for i in range(1, 6)]
            span = (r[:, 5].max() - t0).item() * ns / 1e3
            line += " | %s %3d wgs span %.2f us: prologue %.2f first tile %.2f K loop %.2f stage %.2f epilogue+ack %.2f" % (tag, r.shape[0], span, ph[0], ph[1], ph[2], ph[3], ph[4])
        print(line)
