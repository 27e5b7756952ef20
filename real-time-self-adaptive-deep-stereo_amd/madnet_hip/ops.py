"""Host-side operator layer: torch tensors (storage only) -> C-ABI calls.

Mirrors the operator set of Nets/sharedLayers.py:23-92 and Data_utils/preprocessing.py /
Losses/loss_factory.py hot functions, with explicit forward and gradient entry points (the
reference gets its gradients from TF's autodiff; here every gradient is a hand-written kernel).

All tensors are float32 NHWC.  A `View` is (pointer, B, H, W, C, ld): a channel slice of a
possibly wider buffer, which is how tf.concat is made free.
"""
import ctypes as C
import torch
from . import _ffi


def same_pad(in_size, k, stride=1, dilation=1):
    """TF 'SAME': out=ceil(in/s), pad_total=max((out-1)*s+keff-in,0), before=total//2."""
    keff = (k - 1) * dilation + 1
    out = -(-in_size // stride)
    total = max((out - 1) * stride + keff - in_size, 0)
    return out, total // 2, total - total // 2


class View(object):
    __slots__ = ("t", "ptr", "B", "H", "W", "C", "ld")

    def __init__(self, t, B, H, W, C_, ld, coff=0):
        self.t, self.B, self.H, self.W, self.C, self.ld = t, B, H, W, C_, ld
        self.ptr = t.data_ptr() + 4 * coff

    @property
    def npix(self):
        return self.B * self.H * self.W

    def slice(self, c0, c1):
        v = View(self.t, self.B, self.H, self.W, c1 - c0, self.ld)
        v.ptr = self.ptr + 4 * c0
        return v


def view(t):
    """View of a plain contiguous [B,H,W,C] (or [B,H,W]) float32 tensor."""
    assert t.dtype == torch.float32 and t.is_contiguous(), "float32 contiguous tensors only"
    if t.dim() == 3:
        B, H, W = t.shape
        return View(t, B, H, W, 1, 1)
    B, H, W, Cc = t.shape
    return View(t, B, H, W, Cc, Cc)


def _p(x):
    if x is None:
        return None
    if isinstance(x, View):
        return C.c_void_p(x.ptr)
    if isinstance(x, torch.Tensor):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(x)


def conv_desc(B, Hi, Wi, Ho, Wo, K, N, kh, kw, stride, dil, pad_t, pad_l, mode, w_trans,
              in_ld, out_ld, mask_ld=0, accumulate=0, alpha=1.0, mask_alpha=1.0, mask_c0=0, mask_c1=0, precision=None):
    return _ffi.ConvDesc(B, Hi, Wi, Ho, Wo, K, N, kh, kw, stride, dil, pad_t, pad_l, mode, w_trans,
                         in_ld, out_ld, mask_ld, accumulate, alpha, mask_alpha, mask_c0, mask_c1,
                         PRECISION if precision is None else precision)


# module-wide arithmetic mode of the conv family while a plan is being recorded (mh_conv_desc.precision codes):
#   0 = exact fp32 MFMA (the parity path)      1 = bf16 MFMA operands, fp32 accumulate (throughput)
#   2 = split-bf16 ("bf16x3"): every fp32 operand is carried as hi + lo bf16 and a product costs three bf16 MFMAs
#       (hi*hi + hi*lo + lo*hi, fp32 accumulate): ~2^-16 relative error per product instead of 2^-8.  Kernels without an x3
#       instance run exact fp32 for this code.
# PRECISION applies to the forward convolutions, PRECISION_BWD (None = same) to input and filter gradients.
# Engine precision names -> (forward, backward):
#   'fp32'  (0, 0)   'bf16' (1, 1)
#   'mixed' (2, 1)   forward within the 1e-3 px EPE tolerance of the fp32 oracle, gradients in bf16 (a gradient error moves
#                    the NEXT frame's weights by lr * |dg|; the disparity of the current frame never sees it)
PRECISION = 0
PRECISION_BWD = None
PRECISION_CODES = {"fp32": (0, 0), "bf16": (1, 1), "mixed": (2, 1)}


class precision_scope(object):
    """with precision_scope('mixed'): ... -- sets PRECISION / PRECISION_BWD while a plan is recorded."""

    def __init__(self, name):
        if name not in PRECISION_CODES:
            raise ValueError("precision must be one of %s" % sorted(PRECISION_CODES))
        self.codes = PRECISION_CODES[name]

    def __enter__(self):
        global PRECISION, PRECISION_BWD
        self.saved = (PRECISION, PRECISION_BWD)
        PRECISION, PRECISION_BWD = self.codes
        return self

    def __exit__(self, *a):
        global PRECISION, PRECISION_BWD
        PRECISION, PRECISION_BWD = self.saved
        return False


def _bwd_precision():
    return PRECISION if PRECISION_BWD is None else PRECISION_BWD


def conv_geometry(H, W, kh, kw, stride, dil):
    Ho, pt, _ = same_pad(H, kh, stride, dil)
    Wo, pl, _ = same_pad(W, kw, stride, dil)
    return Ho, Wo, pt, pl


def conv2d_fwd(lib, x, w, b, out, stride=1, dil=1, alpha=1.0, accumulate=False, mask_ref=None, mask_alpha=1.0,
               mask_range=(0, 0), stream=None, precision=None, wb=None, shadow=None, out_planes=None):
    """out (+)= leaky(conv2d_SAME(x, w) + b) [* leaky'(mask_ref)].  x,out: View; w: HWIO [kh,kw,Cin,Cout].
    wb: the layer's MFMA fragment bank (pack_weights) -- split-bf16 3x3 layers then stream their weights from it.
    out_planes: Planes of `out` (hi + lo) the launch writes too (mh_conv2d_sh4: the producer side of conv2d_planes)."""
    kh, kw, cin, cout = w.shape
    Ho, Wo, pt, pl = conv_geometry(x.H, x.W, kh, kw, stride, dil)
    assert (out.H, out.W, out.C) == (Ho, Wo, cout) and x.C == cin
    d = conv_desc(x.B, x.H, x.W, Ho, Wo, cin, cout, kh, kw, stride, dil, pt, pl, 0, 0, x.ld, out.ld, alpha=alpha,
                  mask_ld=(mask_ref.ld if mask_ref is not None else 0), accumulate=int(accumulate), mask_alpha=mask_alpha,
                  mask_c0=mask_range[0], mask_c1=mask_range[1], precision=precision)
    if out_planes is not None:
        assert (out_planes.B, out_planes.H, out_planes.W, out_planes.C) == (out.B, out.H, out.W, out.C)
        lib.conv2d_sh4(C.byref(d), _p(x), _p(w), _p(wb), _p(b), _p(out), _p(mask_ref), C.c_void_p(out_planes.hi.ptr), C.c_void_p(out_planes.lo.ptr), _p(stream))
    elif shadow is not None:     # shadow: ops.Shadow of `out` -- the epilogue also writes bf16(out) there (operand of wgrad_stream)
        assert (shadow.B, shadow.H, shadow.W, shadow.C) == (out.B, out.H, out.W, out.C)
        lib.conv2d_sh(C.byref(d), _p(x), _p(w), _p(wb), _p(b), _p(out), _p(mask_ref), C.c_void_p(shadow.ptr), _p(stream))
    elif wb is not None:
        lib.conv2d_wb(C.byref(d), _p(x), _p(w), _p(wb), _p(b), _p(out), _p(mask_ref), _p(stream))
    else:
        lib.conv2d(C.byref(d), _p(x), _p(w), _p(b), _p(out), _p(mask_ref), _p(stream))


PLANES_WHOLE_MAX = int(__import__("os").environ.get("MH_PLANES_WHOLE_MAX", "8"))      # must equal the library's build (csrc/conv_planes.hip: MH_PLANES_WHOLE_MAX)


def planes_kc16(K):
    """layout rule of the 32x32x16 bank images (mh_planes_kc16; tests assert that the two agree): 0 = whole-K, else the K-chunk in 16-channel steps"""
    k16 = (K + 15) // 16
    return 0 if (k16 <= PLANES_WHOLE_MAX or k16 == 13) else 4


def check_planes_rule(qlib):
    """the bank layout rule lives twice (here for sizing, in the library for the kernels): refuse to run on a library built with another rule"""
    for K in (16, 97, 112, 128, 129, 193, 208, 256, 385, 1025):
        if int(qlib.planes_kc16(K)) != planes_kc16(K):
            raise RuntimeError("libmadnet_hip: mh_planes_kc16(%d) = %d, host rule %d (MH_PLANES_WHOLE_MAX mismatch)" % (K, qlib.planes_kc16(K), planes_kc16(K)))


def _k16_padded(K):
    k16, kc = (K + 15) // 16, planes_kc16(K)
    return (k16 + kc - 1) // kc * kc if kc else k16


def pack_bytes(w, planes=2, trans=0):
    kh, kw, K, N = w.shape
    if trans == 2:          # the 32x32x16 register image of conv2d_planes (hi + lo, or hi only: plain bf16 forward)
        return kh * kw * _k16_padded(K) * ((N + 31) // 32) * 1024 * planes
    if trans == 3:          # ... of conv2d_planes_bwd: one plane, reduction over Cout, columns = Cin
        return kh * kw * _k16_padded(N) * ((K + 31) // 32) * 1024
    if trans:
        K, N = N, K
    return kh * kw * ((K + 31) // 32) * ((N + 15) // 16) * planes * 1024


def pack_weights(lib, pairs, device, keep, stream=None):
    """pairs: [(src HWIO tensor, dst tensor of pack_bytes(src, planes, trans) bytes[, planes = 2[, trans = 0]])] -> every dst = the MFMA
    fragment bank of src (include/madnet_hip.h: mh_pack_weights), ONE launch.  planes 2 = hi + lo (split-bf16 forward), 1 = bf16;
    trans 1 = the bank the input gradient reads (reduction over Cout).  `keep`: list that keeps the device table alive as long as the plan."""
    if not pairs:
        return
    arr = (_ffi.PackSeg * len(pairs))()
    blk = 0
    for i, pr in enumerate(pairs):
        src, dst = pr[0], pr[1]
        planes = pr[2] if len(pr) > 2 else 2
        trans = pr[3] if len(pr) > 3 else 0
        kh, kw, K, N = src.shape
        if trans in (1, 3):
            K, N = N, K
        assert dst.numel() * dst.element_size() >= pack_bytes(src, planes, trans) and dst.data_ptr() % 16 == 0
        assert (trans != 3 or planes == 1)
        arr[i].src, arr[i].dst, arr[i].taps, arr[i].K, arr[i].N = src.data_ptr(), dst.data_ptr(), kh * kw, K, N
        arr[i].planes, arr[i].blk0, arr[i].trans = planes, blk, trans
        arr[i].kc16 = planes_kc16(K) if trans in (2, 3) else 0
        if trans in (2, 3):
            blk += (kh * kw * _k16_padded(K) * ((N + 31) // 32) * 64 + 255) // 256
        else:
            blk += (kh * kw * ((K + 31) // 32) * ((N + 15) // 16) * 64 + 255) // 256
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
    keep.append(table)
    lib.pack_weights(C.c_void_p(table.data_ptr()), len(pairs), blk, _p(stream))


def conv2d_dgrad(lib, dz, w, dx, stride=1, dil=1, accumulate=False, mask_ref=None, mask_alpha=1.0, mask_range=(0, 0),
                 stream=None, wb=None, shadow=None, dz_shadow=None, mask_shadow=None):
    """dx (+)= conv2d_backprop_input(dz, w); optionally fused dx *= leaky'(mask_ref).
    dz: View [B,Ho,Wo,Cout]; dx: View [B,H,W,Cin]; w: HWIO of the forward conv."""
    kh, kw, cin, cout = w.shape
    Ho, Wo, pt, pl = conv_geometry(dx.H, dx.W, kh, kw, stride, dil)
    assert (dz.H, dz.W, dz.C) == (Ho, Wo, cout) and dx.C == cin
    d = conv_desc(dx.B, Ho, Wo, dx.H, dx.W, cout, cin, kh, kw, stride, dil, pt, pl, 1, 1, dz.ld, dx.ld,
                  mask_ld=(mask_ref.ld if mask_ref is not None else 0), accumulate=int(accumulate),
                  alpha=1.0, mask_alpha=mask_alpha, mask_c0=mask_range[0], mask_c1=mask_range[1], precision=_bwd_precision())
    if dz_shadow is not None or mask_shadow is not None:
        # dz_shadow / mask_shadow: ops.Shadow of dz / of mask_ref that a producer already wrote: the patch-staged kernel stages the first instead of
        # converting dz and tests the sign of the second instead of reading the fp32 activation (other kernels ignore both)
        assert dz_shadow is None or (dz_shadow.B, dz_shadow.H, dz_shadow.W, dz_shadow.C) == (dz.B, dz.H, dz.W, dz.C)
        assert mask_shadow is None or (mask_ref is not None and (mask_shadow.B, mask_shadow.H, mask_shadow.W, mask_shadow.C) == (dx.B, dx.H, dx.W, dx.C))
        assert shadow is None or (shadow.B, shadow.H, shadow.W, shadow.C) == (dx.B, dx.H, dx.W, dx.C)
        sp = lambda sh: C.c_void_p(sh.ptr) if sh is not None else None
        lib.conv2d_sh3(C.byref(d), _p(dz), sp(dz_shadow), _p(w), _p(wb), None, _p(dx), _p(mask_ref), sp(mask_shadow), sp(shadow), 0, _p(stream))
    elif shadow is not None:   # shadow: ops.Shadow of dx, written by the epilogue (dx is the next layer's dz operand of wgrad_stream)
        assert (shadow.B, shadow.H, shadow.W, shadow.C) == (dx.B, dx.H, dx.W, dx.C)
        lib.conv2d_sh(C.byref(d), _p(dz), _p(w), _p(wb), None, _p(dx), _p(mask_ref), C.c_void_p(shadow.ptr), _p(stream))
    elif wb is not None:       # wb: pack_weights(trans=1, planes=1) bank of w -- the small-layer bank kernel takes it in the bf16 mode
        lib.conv2d_wb(C.byref(d), _p(dz), _p(w), _p(wb), None, _p(dx), _p(mask_ref), _p(stream))
    else:
        lib.conv2d(C.byref(d), _p(dz), _p(w), None, _p(dx), _p(mask_ref), _p(stream))


def conv2d_wgrad(lib, x, dz, dw, db, stride=1, dil=1, stream=None):
    """dw += conv2d_backprop_filter(x, dz) ; db += sum(dz).  dw: HWIO tensor (pre-zeroed)."""
    kh, kw, cin, cout = dw.shape
    Ho, Wo, pt, pl = conv_geometry(x.H, x.W, kh, kw, stride, dil)
    assert (dz.H, dz.W, dz.C) == (Ho, Wo, cout) and x.C == cin
    d = conv_desc(x.B, x.H, x.W, Ho, Wo, cin, cout, kh, kw, stride, dil, pt, pl, 0, 0, x.ld, dz.ld, precision=_bwd_precision())
    lib.conv2d_wgrad(C.byref(d), _p(x), _p(dz), dz.ld, _p(dw), _p(db), _p(stream))


class WgradWorkspace(object):
    """Arena for the per-split partial filter gradients of one training step + the segment table the single
    reduction launch walks (mh_conv2d_wgrad_partial / mh_wgrad_reduce).  The arena is transient within one plan
    execution, so every plan of an engine shares it (`reset()` at the start of each plan build); chunks are never
    freed or moved because recorded plans hold raw pointers into them."""
    CHUNK = 64 << 20            # floats per arena chunk (256 MiB)

    def __init__(self, device):
        self.device = device
        self.chunks = []
        self.reset()

    def reset(self):
        self.ci, self.off = 0, 0

    def alloc(self, nfloats):
        nfloats = (nfloats + 3) // 4 * 4
        while True:
            if self.ci == len(self.chunks):
                self.chunks.append(torch.empty(max(self.CHUNK, nfloats), dtype=torch.float32, device=self.device))
                self.off = 0
            c = self.chunks[self.ci]
            if self.off + nfloats <= c.numel():
                p = c.data_ptr() + 4 * self.off
                self.off += nfloats
                return p
            self.ci += 1
            self.off = 0

    @property
    def nbytes(self):
        return sum(c.numel() for c in self.chunks) * 4


def conv2d_wgrad_partial(lib, qlib, wsa, segs, x, dz, dw, db, stride=1, dil=1, stream=None, direct_ok=True):
    """Like conv2d_wgrad, but atomic-free: the pixel splits' partial sums go to the arena `wsa` and a segment
    (ws, dst=dw, size, splits) is appended to `segs` for the step's single wgrad_reduce launch.
    `qlib` is the real library (split-count query); `lib` may be a Recorder.  direct_ok=False: dw already receives
    another contribution in this step (shared weights) -- always go through a segment."""
    kh, kw, cin, cout = dw.shape
    Ho, Wo, pt, pl = conv_geometry(x.H, x.W, kh, kw, stride, dil)
    assert (dz.H, dz.W, dz.C) == (Ho, Wo, cout) and x.C == cin
    d = conv_desc(x.B, x.H, x.W, Ho, Wo, cin, cout, kh, kw, stride, dil, pt, pl, 0, 0, x.ld, dz.ld, precision=_bwd_precision())
    splits = C.c_int32(0)
    qlib.conv2d_wgrad_partial(C.byref(d), _p(x), _p(dz), dz.ld, None, C.byref(splits), None, None)
    size = dw.numel()
    if splits.value == 1 and direct_ok and dw.data_ptr() % 16 == 0:
        # a single split IS the gradient: let it store straight into dw (same [tap][K][N] layout), nothing to reduce
        lib.conv2d_wgrad_partial(C.byref(d), _p(x), _p(dz), dz.ld, C.c_void_p(dw.data_ptr()), C.byref(splits), _p(db), _p(stream))
        return
    # with several splits the kernels store the bias partial sums behind the filter partials, ws[splits][size] | [splits][cout] (no float atomics:
    # bit-identical replays); the same reduction launch sums them in split order
    nb = cout * splits.value if (db is not None and splits.value > 1) else 0
    ws = wsa.alloc(size * splits.value + nb)
    lib.conv2d_wgrad_partial(C.byref(d), _p(x), _p(dz), dz.ld, C.c_void_p(ws), C.byref(splits), _p(db), _p(stream))
    segs.append((ws, dw.data_ptr(), size, splits.value))
    if nb:
        segs.append((ws + 4 * size * splits.value, db.data_ptr(), cout, splits.value, 1))      # db += (the step zeroes its gradient ranges first)


def wgrad_reduce(lib, segs, device, keep, stream=None, accumulate=False):
    """One launch that sums the splits of every segment recorded by conv2d_wgrad_partial.  `keep`: list that
    keeps the device table alive as long as the plan.  accumulate: dst += sum (a weight used by a second conv --
    its segments go into a second launch after the first)."""
    if not segs:
        return
    assert len(set(s[1] for s in segs)) == len(segs), "a filter gradient may appear once per reduction"
    arr = (_ffi.WgradSeg * len(segs))()
    blk = 0
    for k, sg in enumerate(segs):
        ws, dst, size, splits = sg[:4]
        # a 5th field = the segment's own accumulate flag (bias partial sums: always dst += sum onto the zeroed gradient range, so that a single-split use
        # of a shared weight -- an atomic addend -- is never overwritten)
        arr[k].ws, arr[k].dst, arr[k].size, arr[k].splits, arr[k].blk0, arr[k].accumulate = ws, dst, size, splits, blk, int(sg[4] if len(sg) > 4 else accumulate)
        blk += (size + 1023) // 1024
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    table = host.to(device)
    keep.append(table)
    lib.wgrad_reduce(C.c_void_p(table.data_ptr()), len(segs), blk, _p(stream))


def shadow_ld(c):
    """channel stride of the bf16 shadow of a C-channel tensor: C rounded up to 32 (the padding channels stay zero)"""
    return (c + 31) // 32 * 32


class Shadow(object):
    """bf16 NHWC copy of an fp32 tensor for the streaming filter-gradient kernel (mh_wgrad_stream): [B,H,W,shadow_ld(C)], pad = 0."""
    __slots__ = ("t", "B", "H", "W", "C", "ld")

    def __init__(self, B, H, W, C_, device):
        self.B, self.H, self.W, self.C, self.ld = B, H, W, C_, shadow_ld(C_)
        self.t = torch.zeros(B, H, W, self.ld, dtype=torch.bfloat16, device=device)

    @property
    def ptr(self):
        return self.t.data_ptr()


class Planes(object):
    """An activation as TWO bf16 NHWC planes, hi = bf16(x) and lo = bf16(x - hi) ([B,H,W,shadow_ld(C)], pad = 0): the operand format of
    conv2d_planes (csrc/conv_planes.hip).  `hi` is an ordinary Shadow -- the one the streamed filter gradient and the input gradients read."""
    __slots__ = ("hi", "lo")

    def __init__(self, hi, device):
        self.hi = hi
        self.lo = Shadow(hi.B, hi.H, hi.W, hi.C, device)

    B = property(lambda self: self.hi.B)
    H = property(lambda self: self.hi.H)
    W = property(lambda self: self.hi.W)
    C = property(lambda self: self.hi.C)
    ld = property(lambda self: self.hi.ld)


def conv2d_planes_ok(qlib, x, w, dil=1, bf16=False, stride=1):
    """does conv2d_planes have an instance for this 'SAME' layer -- stride 1: 3x3 (dilated); stride 2 (round 6): 3x3 / 5x5 on even sizes?  (qlib = the real
    library; bf16: the one-plane form)"""
    kh, kw, cin, cout = w.shape
    if stride == 2:
        if kh != kw or kh not in (3, 5) or dil != 1 or x.H % 2 or x.W % 2:
            return False
        pad = (kh - 2) // 2
        d = conv_desc(x.B, x.H, x.W, x.H // 2, x.W // 2, cin, cout, kh, kw, 2, 1, pad, pad, 0, 0, 0, 0, precision=1 if bf16 else 2)
        return qlib.conv2d_planes_ok(C.byref(d)) == 1
    if (kh, kw) != (3, 3) or stride != 1:
        return False
    d = conv_desc(x.B, x.H, x.W, x.H, x.W, cin, cout, 3, 3, 1, dil, dil, dil, 0, 0, 0, 0, precision=1 if bf16 else 2)
    return qlib.conv2d_planes_ok(C.byref(d)) == 1


def conv2d_planes(lib, xp, w, wb32, b, out=None, out_planes=None, dil=1, alpha=1.0, stream=None, bf16=False, stride=1):
    """leaky(conv2d_SAME(x, w) + b) in split-bf16 from the input's planes `xp` (Planes); results: `out` (fp32 View or None) and / or
    `out_planes` (Planes, or a bare Shadow = hi plane only).  wb32: pack_weights(trans = 2) bank of w.
    bf16: plain bf16 (one MFMA per product) from the hi plane alone -- xp may be a bare Shadow, wb32 the ONE-plane bank (pack planes = 1)."""
    kh, kw, cin, cout = w.shape
    assert xp.C == cin and ((kh, kw) == (3, 3) if stride == 1 else (stride == 2 and kh == kw and dil == 1 and xp.H % 2 == 0 and xp.W % 2 == 0))
    if bf16:
        xhi = xp.hi if isinstance(xp, Planes) else xp
        xp = _HiOnly(xhi)
    Ho, Wo = xp.H // stride, xp.W // stride
    pad = dil if stride == 1 else (kh - 2) // 2                  # TF 'SAME': stride 2 on even sizes pads (k - 2) // 2 in front
    d = conv_desc(xp.B, xp.H, xp.W, Ho, Wo, cin, cout, kh, kw, stride, dil, pad, pad, 0, 0, 0, (out.ld if out is not None else 0), alpha=alpha,
                  precision=1 if bf16 else 2)
    ohi = olo = None
    opld = 0
    if out_planes is not None:
        hi = out_planes.hi if isinstance(out_planes, Planes) else out_planes
        lo = out_planes.lo if isinstance(out_planes, Planes) else None
        assert (hi.B, hi.H, hi.W, hi.C) == (xp.B, Ho, Wo, cout)
        ohi, olo, opld = C.c_void_p(hi.ptr), (C.c_void_p(lo.ptr) if lo is not None else None), hi.ld
    if out is not None:
        assert (out.B, out.H, out.W, out.C) == (xp.B, Ho, Wo, cout)
    lib.conv2d_planes(C.byref(d), C.c_void_p(xp.hi.ptr), (C.c_void_p(xp.lo.ptr) if xp.lo is not None else None), xp.ld, _p(wb32), _p(b), _p(out), ohi, olo, opld,
                      _p(stream))


class _HiOnly(object):
    """a Planes-shaped handle on a bare Shadow (the one-plane forms)"""
    __slots__ = ("hi", "lo")

    def __init__(self, hi):
        self.hi, self.lo = hi, None

    B = property(lambda self: self.hi.B)
    H = property(lambda self: self.hi.H)
    W = property(lambda self: self.hi.W)
    C = property(lambda self: self.hi.C)
    ld = property(lambda self: self.hi.ld)


def conv2d_planes_bwd_ok(qlib, dx, w, dil=1, stride=1):
    """does conv2d_planes_bwd have an instance for the input gradient of this 'SAME' 3x3 layer (w: HWIO of the forward layer; stride 1, or stride 2 on
    even sizes)?"""
    kh, kw, cin, cout = w.shape
    if stride == 2:
        if kh != kw or kh not in (3, 5) or dil != 1 or dx.H % 2 or dx.W % 2:
            return False
        pad = (kh - 2) // 2
        d = conv_desc(dx.B, dx.H, dx.W, dx.H // 2, dx.W // 2, cin, cout, kh, kw, 2, 1, pad, pad, 0, 0, dx.ld, 0, precision=1)
    elif (kh, kw) != (3, 3):
        return False
    else:
        d = conv_desc(dx.B, dx.H, dx.W, dx.H, dx.W, cin, cout, 3, 3, 1, dil, dil, dil, 0, 0, dx.ld, 0, precision=1)
    return qlib.conv2d_planes_bwd_ok(C.byref(d)) == 1


def conv2d_planes_bwd(lib, dz_shadow, w, wb32t, dx=None, dx_shadow=None, mask_shadow=None, mask_alpha=1.0, dil=1, stream=None, mask_range=(0, 0), stride=1, accumulate=False):
    """dx = conv2d_backprop_input(dz, w) * leaky'(mask) from the bf16 shadow of dz (mh_conv2d_planes_bwd): w HWIO [3,3,Cin,Cout] of the forward layer,
    wb32t = pack_weights(trans = 3) bank; results: dx (fp32 View or None) and / or dx_shadow (Shadow); mask_shadow: Shadow of the layer's input."""
    kh, kw, cin, cout = w.shape
    assert dz_shadow.C == cout and ((kh, kw) == (3, 3) or (stride == 2 and kh == kw == 5))
    B, H, W = dz_shadow.B, dz_shadow.H * stride, dz_shadow.W * stride              # size of dx (stride 2: even sizes, 'SAME' pads (k - 2) // 2 in front)
    assert stride in (1, 2) and (stride == 1 or dil == 1)
    pad = dil if stride == 1 else (kh - 2) // 2
    d = conv_desc(B, H, W, dz_shadow.H, dz_shadow.W, cin, cout, kh, kw, stride, dil, pad, pad, 0, 0, (dx.ld if dx is not None else 0), 0, mask_alpha=mask_alpha,
                  precision=1, mask_c0=mask_range[0], mask_c1=mask_range[1], accumulate=int(bool(accumulate)))     # accumulate: (old + new) * mask -- the stride-2 5x5 form only
    for t in (dx, dx_shadow, mask_shadow):
        assert t is None or (t.B, t.H, t.W, t.C) == (B, H, W, cin)
    sp = lambda sh: C.c_void_p(sh.ptr) if sh is not None else None
    lib.conv2d_planes_bwd(C.byref(d), C.c_void_p(dz_shadow.ptr), dz_shadow.ld, _p(wb32t), sp(mask_shadow), (mask_shadow.ld if mask_shadow is not None else 0),
                          _p(dx), sp(dx_shadow), (dx_shadow.ld if dx_shadow is not None else 0), _p(stream))


def plane_split(lib, pairs, device, keep, stream=None):
    """pairs: [(View src, Planes or Shadow dst)] -> hi = bf16(src), lo = bf16(src - hi) (a bare Shadow: hi only), ONE launch (mh_plane_split)."""
    if not pairs:
        return
    arr = (_ffi.PlaneSeg * len(pairs))()
    blk = 0
    for i, (src, dst) in enumerate(pairs):
        hi = dst.hi if isinstance(dst, Planes) else dst
        lo = dst.lo if isinstance(dst, Planes) else None
        src2 = None
        if isinstance(src, (tuple, list)):      # (View a, View b): the planes of tf.concat([a, b], -1)
            src, src2 = src
            assert (src2.B, src2.H, src2.W) == (src.B, src.H, src.W)
        assert (src.B, src.H, src.W, src.C + (src2.C if src2 is not None else 0)) == (hi.B, hi.H, hi.W, hi.C), "planes / source geometry mismatch"
        arr[i].src, arr[i].hi, arr[i].lo, arr[i].npix, arr[i].C = src.ptr, hi.ptr, (lo.ptr if lo is not None else None), src.npix, src.C
        arr[i].src_ld, arr[i].dst_ld, arr[i].blk0 = src.ld, hi.ld, blk
        if src2 is not None:
            arr[i].src2, arr[i].C2, arr[i].src2_ld = src2.ptr, src2.C, src2.ld
        blk += (src.npix * (hi.ld // 8) + 255) // 256
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
    keep.append(table)
    if hasattr(lib, "note_refs"):
        flat = [v for src, _ in pairs for v in (src if isinstance(src, (tuple, list)) else (src,))]
        lib.note_refs([(v.ptr, 4 * v.npix * v.ld) for v in flat])
    lib.plane_split(C.c_void_p(table.data_ptr()), len(pairs), blk, _p(stream))


class InputTable(object):
    """The frame table of mh_fetch_inputs: page-locked host memory the device reads through its device-side address, rewritten by the host between two replays of a
    captured step (after the previous replay has completed) -- no API call, no copy launch in front of the step.  set([tensor | None, ...]): entry k = the device
    tensor the step's k-th input buffer is filled from (uint8 or float32, contiguous), None = leave the buffer as it is."""

    def __init__(self, lib, device):
        cuda = torch.device(device).type == "cuda"
        self.buf = torch.zeros(C.sizeof(_ffi.InputTable), dtype=torch.uint8, pin_memory=cuda)
        self.tab = _ffi.InputTable.from_address(self.buf.data_ptr())
        if cuda:
            dp = C.c_void_p()
            lib.host_device_pointer(C.c_void_p(self.buf.data_ptr()), C.byref(dp))
            self.ptr = dp.value
        else:
            self.ptr = self.buf.data_ptr()
        self.held = []                 # keeps the tensors of the current entries alive

    def set(self, tensors):
        assert len(tensors) <= _ffi.FETCH_MAX
        for k in range(_ffi.FETCH_MAX):
            t = tensors[k] if k < len(tensors) else None
            self.tab.src[k] = t.data_ptr() if t is not None else None
            self.tab.u8[k] = 1 if (t is not None and t.dtype == torch.uint8) else 0
        self.held = [t for t in tensors if t is not None]

    def clear(self):
        self.set([])


def fetch_inputs(lib, table_ptr, dsts, stream=None):
    """the step's first node: every destination tensor (float32, contiguous) <- the tensor the table names for it (mh_fetch_inputs)"""
    n = len(dsts)
    dp = (C.c_void_p * n)(*[d.data_ptr() for d in dsts])
    cnt = (C.c_int64 * n)(*[d.numel() for d in dsts])
    assert all(d.dtype == torch.float32 and d.is_contiguous() for d in dsts)
    lib.fetch_inputs(C.c_void_p(table_ptr), dp, cnt, n, _p(stream))


def shadow_cast(lib, pairs, device, keep, stream=None):
    """pairs: [(View src, Shadow dst)] -> every dst = bf16(src), ONE launch (mh_shadow_cast).  `keep` keeps the device table alive."""
    if not pairs:
        return
    arr = (_ffi.ShadowSeg * len(pairs))()
    blk = 0
    for i, (src, dst) in enumerate(pairs):
        assert (src.B, src.H, src.W, src.C) == (dst.B, dst.H, dst.W, dst.C), "shadow / source geometry mismatch"
        arr[i].src, arr[i].dst, arr[i].npix, arr[i].C = src.ptr, dst.ptr, src.npix, src.C
        arr[i].src_ld, arr[i].dst_ld, arr[i].blk0 = src.ld, dst.ld, blk
        blk += (src.npix * (dst.ld // 8) + 255) // 256
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
    keep.append(table)
    if hasattr(lib, "note_refs"):
        lib.note_refs([(src.ptr, 4 * src.npix * src.ld) for src, _ in pairs])
    lib.shadow_cast(C.c_void_p(table.data_ptr()), len(pairs), blk, _p(stream))


import os as _os
import threading

# serialises plan recording: recordings resolve process-wide tuning hooks (filter-gradient split targets) into the plan, and DispNet's recording
# scopes one of them (RLock: a recording may nest)
TUNE_LOCK = threading.RLock()

# waves per workgroup of the streaming filter-gradient kernel (0 = the caller's choice: the engines take 4 for a batch-1 step -- the
# kernel then shares the chip with the input-gradient chain, 1.870 -> 1.818 ms per step -- and 8 for batched streams: 163 -> 138 us per batch)
WGRAD_STREAM_WAVES = 0            # 0 = the caller's choice (4 waves at batch 1, 8 for batched streams: r03 #4)
WGRAD_STREAM_WGS = 256        # workgroups a batch is divided into (one per CU)


def wgrad_stream_ok(x, dz, dw, stride, dil):
    """layers the streaming kernel covers: 3x3 'SAME', stride 1 (dilation 1 .. 16) or stride 2 (dilation 1, even sizes: padding only behind); the
    channel stride of the input's shadow is its channel count rounded up to 32, so very thin inputs (the 3-channel image) stay on the tiled kernels"""
    kh, kw = dw.shape[0], dw.shape[1]
    if not (kh == 3 and kw == 3 and x.C >= 8):
        return False
    if stride == 1:
        return 1 <= dil <= 16 and (x.H, x.W) == (dz.H, dz.W)
    return stride == 2 and dil == 1 and x.H % 2 == 0 and x.W % 2 == 0 and (x.H, x.W) == (2 * dz.H, 2 * dz.W)


def wgrad_stream(lib, qlib, wsa, segs, items, device, keep, stream=None, target_wgs=None, nwaves=None):
    """Filter + bias gradients of a batch of stride-1 3x3 layers in ONE launch from bf16 shadows (mh_wgrad_stream).
    items: [(Shadow x, Shadow dz, dw tensor (HWIO, fp32), db tensor or None, dil)].  Per-split partial sums go to the arena `wsa` and a
    segment per layer is appended to `segs` for the batch's wgrad_reduce; a layer with ONE split stores straight into dw.
    `qlib` = the real library (host-side planner); `lib` may be a Recorder."""
    if not items:
        return
    # stride-2 layers: an instance of their own; together with stride-1 layers of dilation <= 8 (a pyramid batch) the mixed kernel, ONE launch
    s2 = [it for it in items if it[0].H == 2 * it[1].H]
    mixed = bool(s2) and len(s2) != len(items)
    if mixed and max(it[4] for it in items) > 8:
        wgrad_stream(lib, qlib, wsa, segs, [it for it in items if it[0].H != 2 * it[1].H], device, keep, stream, target_wgs, nwaves)
        items, mixed = s2, False
    stride = 2 if (s2 and not mixed) else 1
    n = len(items)
    arr = (_ffi.WgsLayer * n)()
    max_dil = 1
    for i, (xs, zs, dw, db, dil) in enumerate(items):
        kh, kw, K, N = dw.shape
        st_i = 2 if xs.H == 2 * zs.H else 1
        assert kh == 3 and kw == 3 and (xs.B, xs.H, xs.W) == (zs.B, st_i * zs.H, st_i * zs.W) and xs.C == K and zs.C == N and (mixed or st_i == stride)
        L = arr[i]
        L.x, L.dz, L.db = xs.ptr, zs.ptr, (db.data_ptr() if db is not None else None)
        L.B, L.H, L.W, L.K, L.N, L.dil, L.x_ld, L.dz_ld, L.stride = zs.B, zs.H, zs.W, K, N, dil, xs.ld, zs.ld, st_i
        max_dil = max(max_dil, dil)
    nw = WGRAD_STREAM_WAVES or nwaves or 8
    if max_dil > 8:
        nw = min(nw, 7)                   # 64-pixel row slots: 7 x 20 KB of rings fit the 160 KB LDS
    if stride == 2 or mixed:
        nw, max_dil = min(nw, 5), (-3 if mixed else -2)      # 80-pixel row slots, two new rows per step: 5 x 29 KB of rings
    nblk = C.c_int32(0)
    qlib.wgrad_stream_plan(arr, n, target_wgs or WGRAD_STREAM_WGS, nw, C.byref(nblk))
    for i, (xs, zs, dw, db, dil) in enumerate(items):
        L = arr[i]
        size = dw.numel()
        if L.splits == 1 and dw.data_ptr() % 16 == 0:
            L.ws = dw.data_ptr()
        else:
            # (splits > 1, or a dw that is not 16-byte aligned)  bias partial sums behind the filter partials, summed by the same reduction launch
            nb = L.N * L.splits if (db is not None and L.splits > 1) else 0
            L.ws = wsa.alloc(size * L.splits + nb)
            segs.append((L.ws, dw.data_ptr(), size, L.splits))
            if nb:
                segs.append((L.ws + 4 * size * L.splits, db.data_ptr(), L.N, L.splits, 1))
        if hasattr(lib, "tally_wgrad"):
            lib.tally_wgrad(zs.B, zs.H, zs.W, L.K, L.N, 9, L.splits)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
    keep.append(table)
    lib.wgrad_stream(C.c_void_p(table.data_ptr()), n, nblk.value, nw, max_dil, _p(stream))


def conv2d_transpose_fwd(lib, x, w, b, out, stride=2, alpha=1.0, stream=None, precision=None):
    """tf.nn.conv2d_transpose(x, w[kh,kw,Cout,Cin], 'SAME') + b, leaky (sharedLayers.py:80-92):
    the input-gradient of a SAME conv with HWIO = [kh,kw,I=Cout,O=Cin]."""
    kh, kw, cout, cin = w.shape
    Ho, Wo = x.H * stride, x.W * stride
    _, _, pt, pl = conv_geometry(Ho, Wo, kh, kw, stride, 1)
    assert (out.H, out.W, out.C) == (Ho, Wo, cout) and x.C == cin
    d = conv_desc(x.B, x.H, x.W, Ho, Wo, cin, cout, kh, kw, stride, 1, pt, pl, 1, 1, x.ld, out.ld, alpha=alpha, precision=precision)
    lib.conv2d(C.byref(d), _p(x), _p(w), _p(b), _p(out), None, _p(stream))


def corr_fwd(lib, L, R, out, max_disp, stride=1, coff=0, u=None, copy_left=False, zero_tail=False, stream=None, precision=None):
    """precision: None = the forward code of the plan being recorded (PRECISION); only the large-D MFMA kernel uses it."""
    prec = PRECISION if precision is None else precision
    if prec == 0:
        lib.corr_fwd(_p(L), L.ld, _p(R), R.ld, _p(u), _p(out), out.ld, coff, L.B, L.H, L.W, L.C, max_disp, stride,
                     int(copy_left), int(zero_tail), _p(stream))
    else:
        lib.corr_fwd_prec(_p(L), L.ld, _p(R), R.ld, _p(u), _p(out), out.ld, coff, L.B, L.H, L.W, L.C, max_disp, stride,
                          int(copy_left), int(zero_tail), prec, _p(stream))


def level_front_fwd(lib, Vc, mul, L, R, out, Rw, u, max_disp, coff, zero_tail=True, stream=None, planes=None):
    """u = mul * resize(Vc) ; Rw = warp(R, u) ; out = [L | corr(L, Rw) | u | 0]  -- one launch (mh_level_front_fwd).
    Vc: [B,H,Wc] tensor; L, R, Rw: Views [B,H,W,C]; out: View of the estimator input; u: [B,H,W] tensor.
    planes: Planes (hi + lo) or Shadow (hi only) of `out` that the launch writes too."""
    if planes is None:
        lib.level_front_fwd(_p(Vc), Vc.shape[1], Vc.shape[2], mul, _p(L), L.ld, _p(R), R.ld, _p(out), out.ld, coff, _p(Rw), Rw.ld, _p(u),
                            L.B, L.H, L.W, L.C, max_disp, int(zero_tail), _p(stream))
        return
    hi = planes.hi if isinstance(planes, Planes) else planes
    lo = planes.lo if isinstance(planes, Planes) else None
    assert (hi.B, hi.H, hi.W) == (L.B, L.H, L.W) and hi.C == coff + 2 * max_disp + 2 and hi.ld >= hi.C
    lib.level_front_fwd_planes(_p(Vc), Vc.shape[1], Vc.shape[2], mul, _p(L), L.ld, _p(R), R.ld, _p(out), out.ld, coff, _p(Rw), Rw.ld, _p(u),
                               L.B, L.H, L.W, L.C, max_disp, int(zero_tail), C.c_void_p(hi.ptr), (C.c_void_p(lo.ptr) if lo is not None else None), hi.ld, _p(stream))


def level_front_head_fwd(lib, X, hw, hb, Vc, mul, L, R, out, Rw, u, max_disp, coff, zero_tail=True, stream=None, planes=None):
    """level_front_fwd whose coarse disparity Vc is COMPUTED in the launch: Vc = conv3x3(X, hw) + hb, the disparity head of the coarser level (mh_level_front_head_fwd).
    X: View [B,Hc,Wc,K]; hw: the head's [3,3,K,1] weights; hb: its bias tensor or None; Vc: [B,Hc,Wc] tensor (output)."""
    hi = lo = None
    if planes is not None:
        hi = planes.hi if isinstance(planes, Planes) else planes
        lo = planes.lo if isinstance(planes, Planes) else None
        assert (hi.B, hi.H, hi.W) == (L.B, L.H, L.W) and hi.C == coff + 2 * max_disp + 2 and hi.ld >= hi.C
    assert (X.B, X.H, X.W) == (L.B, Vc.shape[1], Vc.shape[2]) and tuple(hw.shape) == (3, 3, X.C, 1)
    lib.level_front_head_fwd(_p(X), X.ld, X.C, _p(hw), _p(hb), _p(Vc), Vc.shape[1], Vc.shape[2], mul, _p(L), L.ld, _p(R), R.ld, _p(out), out.ld, coff, _p(Rw), Rw.ld,
                             _p(u), L.B, L.H, L.W, L.C, max_disp, int(zero_tail), (C.c_void_p(hi.ptr) if hi is not None else None),
                             (C.c_void_p(lo.ptr) if lo is not None else None), (hi.ld if hi is not None else 0), _p(stream))


def corr_bwd(lib, g, L, R, dL, dR, max_disp, stride=1, coff=0, du=None, acc_l=False, acc_r=False, acc_u=False,
             copy_left=False, stream=None, precision=None):
    """precision: None = the backward code of the plan being recorded; only the large-D MFMA kernels use it (1 = bf16 operands)."""
    prec = _bwd_precision() if precision is None else precision
    lib.corr_bwd_prec(_p(g), g.ld, coff, _p(L), L.ld, _p(R), R.ld, _p(dL), dL.ld, int(acc_l), _p(dR), dR.ld, int(acc_r),
                      _p(du), int(acc_u), L.B, L.H, L.W, L.C, max_disp, stride, int(copy_left), int(prec), _p(stream))


def corr_warp_bwd(lib, g, L, Rw, img, u, dL, dimg, du, max_disp, stride=1, coff=0, acc_l=False, copy_left=False, stream=None, acc_img=True):
    """corr_bwd(g, L, Rw -> dL, dRw, du = g[disparity channel]) + warp_bwd(dRw, img, u -> dimg scatter, du += coordinate gradient) in one
    launch; dRw is never stored.  acc_img=True: dimg holds zeros / earlier contributions and is added to; False: this launch is dimg's first writer
    (overwritten, nothing read, no zero fill needed).  dimg or du may be None."""
    lib.corr_warp_bwd(_p(g), g.ld, coff, _p(L), L.ld, _p(Rw), Rw.ld, _p(img), img.ld, _p(u), _p(dL), dL.ld, int(bool(acc_l)) | (0 if acc_img else 2),
                      _p(dimg), dimg.ld if dimg is not None else img.ld, _p(du), L.B, L.H, L.W, L.C, max_disp, stride, int(copy_left), _p(stream))


def warp_fwd(lib, img, u, out, stream=None):
    lib.warp_fwd(_p(img), img.ld, _p(u), _p(out), out.ld, img.B, img.H, img.W, img.C, _p(stream))


def warp_bwd(lib, g, img, u, dimg, du=None, acc_u=False, stream=None):
    """dimg None: coordinate gradient only; du None: scatter only."""
    lib.warp_bwd(_p(g), g.ld, _p(img), img.ld, _p(u), _p(dimg), dimg.ld if dimg is not None else img.ld, _p(du), int(acc_u),
                 img.B, img.H, img.W, img.C, _p(stream))


def resize_fwd(lib, x, out, Hr, Wr, cy=0, cx=0, mul=1.0, mode=0, stream=None):
    """x: [B,Hi,Wi] tensor; out: [B,Ho,Wo] tensor (crop of the virtual [Hr,Wr] resize at (cy,cx))."""
    B, Hi, Wi = x.shape[0], x.shape[1], x.shape[2]
    Ho, Wo = out.shape[1], out.shape[2]
    lib.resize_fwd(_p(x), _p(out), B, Hi, Wi, Hr, Wr, cy, cx, Ho, Wo, mul, mode, _p(stream))


def resize_bwd(lib, g, x, dx, Hr, Wr, cy=0, cx=0, mul=1.0, mode=0, accumulate=False, stream=None):
    B, Hi, Wi = x.shape[0], x.shape[1], x.shape[2]
    Ho, Wo = g.shape[1], g.shape[2]
    lib.resize_bwd(_p(g), _p(x), _p(dx), int(accumulate), B, Hi, Wi, Hr, Wr, cy, cx, Ho, Wo, mul, mode, _p(stream))


def conv2d_head(lib, x, w, b, out, copies=(), alpha=1.0, stream=None):
    """Forward pass of a disparity head (3x3 Cin -> 1, mh_conv2d_head): out = conv(x, w) + b, ALSO stored at up to two more Views with C = 1
    (`copies`: a channel slot of a concatenated buffer, the buffer the next stage accumulates into)."""
    kh, kw, cin, cout = w.shape
    assert cout == 1 and x.C == cin and len(copies) <= 2 and all((c.B, c.H, c.W, c.C) == (out.B, out.H, out.W, 1) for c in copies)
    Ho, Wo, pt, pl = conv_geometry(x.H, x.W, kh, kw, 1, 1)
    d = conv_desc(x.B, x.H, x.W, Ho, Wo, cin, 1, kh, kw, 1, 1, pt, pl, 0, 0, x.ld, out.ld, alpha=alpha)
    c = list(copies) + [None, None]
    lib.conv2d_head(C.byref(d), _p(x), _p(w), _p(b), _p(out), _p(c[0]), (c[0].ld if c[0] is not None else 0),
                    _p(c[1]), (c[1].ld if c[1] is not None else 0), _p(stream))


def head_bwd(lib, w, dV, dx, mask_ref=None, mask_alpha=1.0, accumulate_dx=False, du=None, Hr=0, Wr=0, mul=1.0, addends=(), dV_shadow=None,
             dx_shadow=None, stream=None):
    """Backward front end of a disparity head (mh_head_bwd): dV [B,H,W] tensor <- either the resize gradient of the finer level's coordinate
    gradient `du` ([B,Ho,Wo] tensor; Hr, Wr, mul as resize_bwd mode 0) or the sum of up to two `addends` (Views with C = 1: channel slices are
    fine); dx (View [B,H,W,N]) (+)= conv2d_backprop_input(dV, w) * leaky'(mask_ref).  w: HWIO [3,3,N,1] of the head conv.  Shadows: ops.Shadow of
    dV (C = 1) / dx, written on the way."""
    B, H, W = dV.shape
    assert tuple(w.shape) == (3, 3, dx.C, 1) and (dx.B, dx.H, dx.W) == (B, H, W)
    d = _ffi.HeadBwdDesc()
    d.B, d.H, d.W, d.N = B, H, W, dx.C
    d.dx_ld, d.mask_ld, d.accumulate_dx, d.mask_alpha = dx.ld, (mask_ref.ld if mask_ref is not None else 0), int(accumulate_dx), mask_alpha
    if du is not None:
        d.kind, d.Hr, d.Wr, d.cy, d.cx, d.Ho, d.Wo, d.mul = 0, Hr, Wr, 0, 0, du.shape[1], du.shape[2], mul
        src0, src1 = _p(du), None
    else:
        assert 1 <= len(addends) <= 2 and all((a.B, a.H, a.W, a.C) == (B, H, W, 1) for a in addends)
        d.kind = 1
        d.src0_ld = addends[0].ld
        d.src1_ld = addends[1].ld if len(addends) > 1 else 0
        src0, src1 = _p(addends[0]), (_p(addends[1]) if len(addends) > 1 else None)
    for sh, c in ((dV_shadow, 1), (dx_shadow, dx.C)):
        assert sh is None or (sh.B, sh.H, sh.W, sh.C) == (B, H, W, c)
    lib.head_bwd(C.byref(d), src0, src1, _p(dV), (C.c_void_p(dV_shadow.ptr) if dV_shadow is not None else None), _p(w), _p(dx), _p(mask_ref),
                 (C.c_void_p(dx_shadow.ptr) if dx_shadow is not None else None), _p(stream))


def resize_image(lib, x, out, stream=None):
    """TF1-legacy bilinear resize of an NHWC image tensor [B,H,W,C] into out [B,Ho,Wo,C] (scale_tensor on the frames)."""
    B, H, W, Cc = x.shape
    lib.resize_image_fwd(_p(x), _p(out), B, H, W, Cc, out.shape[1], out.shape[2], _p(stream))


def conv_image_fwd(lib, frames, Hp, Wp, reflect_t, reflect_l, w, bias, out, stride=2, alpha=1.0, div=1.0, sub=0.0, shadow=None, stream=None):
    """out = leaky(conv3x3(reflect_pad(frames / div - sub), w) + bias) from the frames themselves (mh_conv_image_fwd): frames [NB,H0,W0,3] tensor, out a View
    [NB,ceil(Hp/stride),ceil(Wp/stride),16]; shadow: the Shadow of `out` the launch writes too."""
    NB, H0, W0, Cc = frames.shape
    kh, kw, cin, cout = w.shape
    assert (kh, kw, cin) == (3, 3, Cc) and cout == out.C
    Ho, Wo, pt, pl = conv_geometry(Hp, Wp, 3, 3, stride, 1)
    assert (out.B, out.H, out.W) == (NB, Ho, Wo)
    lib.conv_image_fwd(_p(frames), NB, H0, W0, Cc, Hp, Wp, reflect_t, reflect_l, div, sub, _p(w), _p(bias), cout, stride, pt, pl, alpha, _p(out), out.ld,
                       (C.c_void_p(shadow.ptr) if shadow is not None else None), (shadow.ld if shadow is not None else 0), _p(stream))


def pad_reflect(lib, x, out, pad_t, pad_l, div=1.0, sub=0.0, stream=None):
    """out = reflect_pad(x / div - sub).  x: [B,H,W,C] tensor; out: [B,Hp,Wp,out_ld] tensor."""
    B, H, W, Cc = x.shape
    lib.pad_reflect(_p(x), _p(out), B, H, W, Cc, out.shape[1], out.shape[2], pad_t, pad_l, out.shape[3], div, sub, _p(stream))


def bias_grad_partial(lib, qlib, wsa, segs, dz, db, stream=None):
    """db += column sums of dz without float atomics: per-workgroup partial sums into the arena + one more segment for the batch's wgrad_reduce
    (fixed summation order -- replays of a step are bit-identical).  `qlib` = the real library (grid query); `lib` may be a Recorder."""
    nb = qlib.bias_grad_blocks(dz.npix, dz.C)
    ws = wsa.alloc(nb * dz.C)
    lib.bias_grad_partial(_p(dz), dz.ld, dz.npix, dz.C, C.c_void_p(ws), nb, _p(stream))
    segs.append((ws, db.data_ptr(), dz.C, nb, 1))


def bias_grad(lib, dz, db, stream=None):
    """db += column sums of the View dz."""
    lib.bias_grad(_p(dz), dz.ld, dz.npix, dz.C, _p(db), _p(stream))


def reprojection_loss(lib, left, right, disp, ws, result, ddisp=None, grad_scale=1.0, stream=None, phase=0):
    """phase 0 = everything; 1 = warp + SSIM maps + gradient (what the backward pass waits for); 2 = the reduction of the loss value."""
    B, H, W = disp.shape[0], disp.shape[1], disp.shape[2]
    if phase == 0:
        lib.reprojection_loss(_p(left), _p(right), _p(disp), _p(ws), _p(result), _p(ddisp), grad_scale, B, H, W, _p(stream))
    else:
        lib.reprojection_loss_phase(_p(left), _p(right), _p(disp), _p(ws), _p(result), _p(ddisp), grad_scale, B, H, W, phase, _p(stream))


def proxy_loss(lib, pred, proxy, ws, result, dpred=None, weight=0.01, grad_scale=1.0, stream=None):
    """result[0] = weight * mean_l1(pred, proxy, valid) (loss_factory.get_proxy_loss); dpred = its gradient (optional)."""
    B, H, W = pred.shape[0], pred.shape[1], pred.shape[2]
    lib.proxy_loss(_p(pred), _p(proxy), _p(ws), _p(result), _p(dpred), weight, grad_scale, B, H, W, _p(stream))


def supervised_loss(lib, pred, target, ws, result, dpred=None, weight=1.0, grad_scale=1.0, max_disp=192.0, stream=None):
    """result[0] = weight * mean_l1(pred, target, valid = !(target == 0 | target >= max_disp)) -- one scale of
    loss_factory.get_supervised_loss (Train.py:100); dpred = its gradient (optional).  Workspace as proxy_loss."""
    B, H, W = pred.shape[0], pred.shape[1], pred.shape[2]
    lib.supervised_loss(_p(pred), _p(target), _p(ws), _p(result), _p(dpred), weight, grad_scale, max_disp, B, H, W, _p(stream))


def adam(lib, var, m, v, grad, state, lr, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, n=None, stream=None):
    """tf.train.AdamOptimizer apply; state = device float[2] {beta1_power, beta2_power} (see adam_advance)."""
    lib.adam(_p(var), _p(m), _p(v), _p(grad), n if n is not None else var.numel(), _p(state), lr, beta1, beta2, eps, grad_scale, _p(stream))


def adam_advance(lib, state, beta1=0.9, beta2=0.999, stream=None):
    lib.adam_advance(_p(state), beta1, beta2, _p(stream))


def metrics(lib, disp, gt, ws, result, pixel_th=3.0, stream=None):
    B, H, W = disp.shape[0], disp.shape[1], disp.shape[2]
    lib.metrics(_p(disp), _p(gt), _p(ws), _p(result), pixel_th, B, H, W, _p(stream))


def momentum(lib, var, accum, grad, lr, mom=0.9, grad_scale=1.0, n=None, stream=None):
    lib.momentum(_p(var), _p(accum), _p(grad), n if n is not None else var.numel(), lr, mom, grad_scale, _p(stream))


def copy_channels(lib, src, dst, nch=None, scale=1.0, accumulate=False, stream=None):
    lib.copy_channels(_p(src), src.ld, _p(dst), dst.ld, src.npix, nch if nch is not None else src.C, scale,
                      int(accumulate), _p(stream))


def leaky_bwd(lib, dy, y, alpha, stream=None):
    lib.leaky_bwd(_p(dy), dy.ld, _p(y), y.ld, dy.npix, dy.C, alpha, _p(stream))


def stamp(lib, slots, index, stream=None):
    """slots[index] (int64 device tensor) = the device wall clock when this op runs (mh_stamp)"""
    assert slots.dtype == torch.int64
    lib.stamp(C.c_void_p(slots.data_ptr() + 8 * index), _p(stream))
