#!/bin/bash
TAG=${1:-r02d}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_parity.py -m gpu -q -k level_front 2>&1 | grep -v "^  File" | tail -40 > $OUT/pytest_front.txt
timeout 300 python - > $OUT/front_timing.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "real-time-self-adaptive-deep-stereo_amd")
import torch
from madnet_hip import _ffi, ops
from madnet_hip.benchtools import _time_ms
lib = _ffi.lib(); st = torch.cuda.Stream(); sh = st.cuda_stream
for (B, H, W, C) in [(1, 12, 40, 128), (1, 24, 80, 96), (1, 48, 160, 64), (1, 96, 320, 32)]:
    Vc = torch.randn(B, H // 2, W // 2, device="cuda"); L = torch.randn(B, H, W, C, device="cuda"); R = torch.randn(B, H, W, C, device="cuda")
    ld = (C + 6 + 3) // 4 * 4
    out = torch.empty(B, H, W, ld, device="cuda"); Rw = torch.empty(B, H, W, C, device="cuda"); u = torch.empty(B, H, W, device="cuda")
    ov = ops.View(out, B, H, W, ld, ld)
    with torch.cuda.stream(st):
        def chain():
            ops.resize_fwd(lib, Vc, u, H, W, mul=2.5, mode=0, stream=sh)
            ops.warp_fwd(lib, ops.view(R), u, ops.view(Rw), stream=sh)
            ops.corr_fwd(lib, ops.view(L), ops.view(Rw), ov, 2, 1, coff=C, u=u, copy_left=True, zero_tail=True, stream=sh)
        def fused():
            ops.level_front_fwd(lib, Vc, 2.5, ops.view(L), ops.view(R), ov, ops.view(Rw), u, 2, coff=C, stream=sh)
        t0 = _time_ms(lib, st, chain, 20) * 1e3; t1 = _time_ms(lib, st, fused, 20) * 1e3
    print("%dx%dx%d: chain %.1f us  fused %.1f us  (%s)" % (H, W, C, t0, t1, lib.last_kernel().decode()))
PY
cat $OUT/pytest_front.txt | tail -30; cat $OUT/front_timing.txt
