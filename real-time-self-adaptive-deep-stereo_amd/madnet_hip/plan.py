"""Plan recording / replay: the host compiles a network ONCE into an array of mh_op records
(include/madnet_hip.h); libmadnet_hip.so's native executor replays it with one FFI call, or
captures it into a hipGraph.  This replaces TF1's graph + Session.run machinery for the path
(Stereo_Online_Adaptation.py:208) -- there is no tracing compiler.

`Recorder` exposes the same call surface as `_ffi.Lib`, so the wrappers in ops.py are reused
verbatim to *record* instead of *launch*.  The packing below must match run_op() in
csrc/lib.hip.
"""
import ctypes as C
from . import _ffi


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, C.c_void_p):
        return x.value
    return int(x)


class Recorder(object):
    def __init__(self):
        self.ops = []
        self.keep = []          # python objects (tensors) that must outlive the plan
        # algorithmic work of the recorded step (SURVEY 8(d) definitions): conv = forward + input-gradient launches,
        # wgrad = filter gradients; *_bytes = each operand read / result written once; wgrad_ws_bytes = split-K partial sums
        self.stats = {"conv_flops": 0.0, "conv_bytes": 0.0, "wgrad_flops": 0.0, "wgrad_bytes": 0.0, "wgrad_ws_bytes": 0.0,
                      "grad_bytes": 0.0, "conv_launches": 0, "wgrad_launches": 0}
        self.lane = 0           # scheduling lane of the ops recorded next (mh_op.i[26], include/madnet_hip.h)
        self.join_next = False  # next op: lane 0 first waits for the side lanes
        self.join_lanes_next = 0  # next op: lane 0 first waits for exactly the side lanes of this bit mask (bit l = lane l)
        self.nodefer = False      # side-lane ops recorded now are launched at once (MH_OP_NODEFER)
        self.wgrad_group_max_m = 0  # grouped filter-gradient launches: pixel cap of this plan's layers (0 = library default)
        self.work = {}              # op index -> (algorithmic flops, bytes) of the multi-layer ops (a streamed filter-gradient batch)
        self._pending_work = [0.0, 0.0]
        self.cuts = []              # op indices where compile_parts() splits the recording (a collective goes between the parts)
        self.refs = []              # (op index, data pointer, bytes): tensors an op reaches through a DEVICE TABLE (segment tables of casts / splits /
                                    # streamed filter gradients) -- invisible in the op's own pointer fields, seen by the dead-store post-passes

    def note_refs(self, pairs):
        """pairs: [(data pointer, bytes)] read or written through the device table of the op recorded NEXT"""
        self.refs += [(len(self.ops), int(a), int(nb)) for a, nb in pairs if a]

    def cut(self):
        """Mark a split point: every op recorded so far forms one part (mh_plan_run joins the side lanes at the end of a plan, so
        a part ends with all of its work ordered before whatever the caller enqueues next on the stream)."""
        if len(self.ops) and (not self.cuts or self.cuts[-1] != len(self.ops)):
            self.cuts.append(len(self.ops))

    # -- helpers ---------------------------------------------------------------------------
    def _op(self, kind, ints=(), floats=(), ptrs=(), n=0):
        o = _ffi.Op()
        o.kind = kind
        for k, v in enumerate(ints):
            o.i[k] = int(v)
        for k, v in enumerate(floats):
            o.f[k] = float(v)
        for k, v in enumerate(ptrs):
            o.p[k] = _ptr(v)
        o.n = int(n)
        o.i[26] = (self.lane | (_ffi.OP_JOIN if self.join_next else 0) | ((self.join_lanes_next & 0xff) << 16)
                   | (_ffi.OP_NODEFER if (self.nodefer and self.lane > 0) else 0))
        self.join_next = False
        self.join_lanes_next = 0
        self.ops.append(o)

    @staticmethod
    def _desc_ints(d):
        return [d.B, d.Hi, d.Wi, d.Ho, d.Wo, d.K, d.N, d.kh, d.kw, d.stride, d.dil, d.pad_t, d.pad_l,
                d.mode, d.w_trans, d.in_ld, d.out_ld, d.mask_ld, d.accumulate, d.mask_c0, d.mask_c1]

    # -- same names / argument order as _ffi.Lib (minus the 'mh_' prefix) --------------------
    def _tally(self, d, kind, splits=0):
        taps = d.kh * d.kw
        if kind == "conv":          # mode 0: in = x, out = y ; mode 1: in = dz (Hi x Wi), out = dx
            flops = 2.0 * d.B * (d.Ho * d.Wo if d.mode == 0 else d.Hi * d.Wi) * taps * d.K * d.N
            byts = 4.0 * (d.B * d.Hi * d.Wi * d.K + d.B * d.Ho * d.Wo * d.N + taps * d.K * d.N)
            self.stats["conv_flops"] += flops; self.stats["conv_bytes"] += byts; self.stats["conv_launches"] += 1
        else:
            flops = 2.0 * d.B * d.Ho * d.Wo * taps * d.K * d.N
            byts = 4.0 * (d.B * d.Hi * d.Wi * d.K + d.B * d.Ho * d.Wo * d.N + taps * d.K * d.N)
            self.stats["wgrad_flops"] += flops; self.stats["wgrad_bytes"] += byts; self.stats["wgrad_launches"] += 1
            self.stats["grad_bytes"] += 4.0 * taps * d.K * d.N
            if splits > 1:
                self.stats["wgrad_ws_bytes"] += 2 * 4.0 * splits * taps * d.K * d.N      # written by the splits, read by the reduction

    def conv2d_wb(self, dref, inp, w, wb, bias, out, mask, stream):
        d = dref._obj
        self._tally(d, "conv")
        self._op(_ffi.OP_CONV, self._desc_ints(d) + [0, d.precision], [d.alpha, d.mask_alpha], [inp, w, bias, out, mask, None, wb])

    def conv2d_sh(self, dref, inp, w, wb, bias, out, mask, shadow, stream):
        d = dref._obj
        self._tally(d, "conv")
        self._op(_ffi.OP_CONV, self._desc_ints(d) + [0, d.precision], [d.alpha, d.mask_alpha], [inp, w, bias, out, mask, None, wb, shadow])

    def conv2d_sh4(self, dref, inp, w, wb, bias, out, mask, out_hi, out_lo, stream):
        d = dref._obj
        self._tally(d, "conv")
        self._op(_ffi.OP_CONV, self._desc_ints(d) + [0, d.precision, 32], [d.alpha, d.mask_alpha], [inp, w, bias, out, mask, out_lo, wb, out_hi])

    def conv2d_sh2(self, dref, inp, in_shadow, w, wb, bias, out, mask, shadow, stream):
        d = dref._obj
        self._tally(d, "conv")
        self._op(_ffi.OP_CONV, self._desc_ints(d) + [0, d.precision, 1], [d.alpha, d.mask_alpha], [inp, w, bias, out, mask, in_shadow, wb, shadow])

    def conv2d_sh3(self, dref, inp, in_shadow, w, wb, bias, out, mask, mask_shadow, shadow, flags, stream):
        d = dref._obj
        assert bias is None
        self._tally(d, "conv")
        bits = (1 if in_shadow is not None else 0) | (2 if mask_shadow is not None else 0) | (4 if flags & 1 else 0)
        self._op(_ffi.OP_CONV, self._desc_ints(d) + [0, d.precision, bits], [d.alpha, d.mask_alpha], [inp, w, mask_shadow, out, mask, in_shadow, wb, shadow])

    def conv2d_planes(self, dref, in_hi, in_lo, in_pld, wb32, bias, out, out_hi, out_lo, out_pld, stream):
        d = dref._obj
        self._tally(d, "conv")
        ints = self._desc_ints(d) + [0, (1 if d.precision == 1 else 2), in_pld, out_pld]
        self._op(_ffi.OP_CONV_PLANES, ints, [d.alpha, d.mask_alpha], [in_hi, in_lo, wb32, bias, out, out_hi, out_lo])

    def stamp(self, slot, stream):
        self._op(_ffi.OP_STAMP, [], [], [slot])

    def det_flush(self, dst, twin, n, stream):
        self._op(_ffi.OP_DET_FLUSH, [], [], [dst, twin], n=n)

    def conv2d_planes_bwd(self, dref, dz_hi, dz_pld, wb32t, mask_hi, mask_pld, dx, dx_hi, dx_pld, stream):
        d = dref._obj
        # (tallied as the input-gradient launch it is: flops of the layer, dz in, dx out)
        flops = 2.0 * d.B * d.Hi * d.Wi * 9 * d.K * d.N
        self.stats["conv_flops"] += flops; self.stats["conv_bytes"] += 4.0 * (d.B * d.Hi * d.Wi * (d.K + d.N) + 9 * d.K * d.N); self.stats["conv_launches"] += 1
        ints = self._desc_ints(d) + [0, 1, dz_pld, mask_pld, dx_pld]
        self._op(_ffi.OP_CONV_PLANES_BWD, ints, [d.alpha, d.mask_alpha], [dz_hi, wb32t, mask_hi, dx, dx_hi])

    def plane_split(self, segs, nseg, nblocks, stream):
        self._op(_ffi.OP_PLANE_SPLIT, [nseg, nblocks], [], [segs])

    def pack_weights(self, segs, nseg, nblocks, stream):
        self._op(_ffi.OP_PACK_W, [nseg, nblocks], [], [segs])

    def conv2d(self, dref, inp, w, bias, out, mask, stream):
        d = dref._obj
        self._tally(d, "conv")
        self._op(_ffi.OP_CONV, self._desc_ints(d) + [0, d.precision], [d.alpha, d.mask_alpha], [inp, w, bias, out, mask])

    def conv2d_wgrad(self, dref, inp, dout, dout_ld, dw, db, stream):
        d = dref._obj
        ints = self._desc_ints(d) + [dout_ld, d.precision]
        self._tally(d, "wgrad")
        self._op(_ffi.OP_WGRAD, ints, [d.alpha, d.mask_alpha], [inp, dout, dw, db])

    def conv2d_wgrad_partial(self, dref, inp, dout, dout_ld, ws, splits_ref, db, stream):
        d = dref._obj
        ints = self._desc_ints(d) + [dout_ld, d.precision, splits_ref._obj.value, self.wgrad_group_max_m]
        self._tally(d, "wgrad", splits_ref._obj.value if ws is not None else 0)
        self._op(_ffi.OP_WGRAD_PARTIAL, ints, [d.alpha, d.mask_alpha], [inp, dout, ws, db])

    def shadow_cast(self, segs, nseg, nblocks, stream):
        self._op(_ffi.OP_SHADOW_CAST, [nseg, nblocks], [], [segs])

    def wgrad_stream(self, layers, nlayers, nblocks, nwaves, max_dil, stream):
        self.work[len(self.ops)] = tuple(self._pending_work)
        self._pending_work = [0.0, 0.0]
        self._op(_ffi.OP_WGRAD_STREAM, [nlayers, nblocks, nwaves, max_dil], [], [layers])

    def tally_wgrad(self, B, H, W, K, N, taps, splits):
        """work of one layer of a streamed filter-gradient batch (the batch is ONE op)"""
        self.stats["wgrad_flops"] += 2.0 * B * H * W * taps * K * N
        self.stats["wgrad_bytes"] += 4.0 * (B * H * W * (K + N) + taps * K * N)
        self._pending_work[0] += 2.0 * B * H * W * taps * K * N
        self._pending_work[1] += 2.0 * B * H * W * ((K + 31) // 32 * 32 + (N + 31) // 32 * 32) + 4.0 * taps * K * N      # bf16 shadows in, fp32 gradient out
        self.stats["wgrad_launches"] += 1
        self.stats["grad_bytes"] += 4.0 * taps * K * N
        if splits > 1:
            self.stats["wgrad_ws_bytes"] += 2 * 4.0 * splits * taps * K * N

    def conv2d_head(self, dref, inp, w, bias, out, out2, out2_ld, out3, out3_ld, stream):
        d = dref._obj
        self._tally(d, "conv")
        self._op(_ffi.OP_HEAD_FWD, self._desc_ints(d) + [0, d.precision, out2_ld, out3_ld], [d.alpha, d.mask_alpha], [inp, w, bias, out, out2, out3])

    def head_bwd(self, dref, src0, src1, dV, dV_shadow, w, dx, mask, dx_shadow, stream):
        d = dref._obj
        self._op(_ffi.OP_HEAD_BWD, [d.kind, d.B, d.H, d.W, d.N, d.Hr, d.Wr, d.cy, d.cx, d.Ho, d.Wo, d.src0_ld, d.src1_ld, d.dx_ld, d.mask_ld,
                                    d.accumulate_dx], [d.mul, d.mask_alpha], [src0, src1, dV, dV_shadow, w, dx, mask, dx_shadow])

    def wgrad_reduce(self, segs, nseg, nblocks, stream):
        self._op(_ffi.OP_WGRAD_REDUCE, [nseg, nblocks], [], [segs])

    def corr_fwd(self, L, l_ld, R, r_ld, u, out, out_ld, coff, B, H, W, Cc, md, stride, copy_left, zero_tail, stream):
        self._op(_ffi.OP_CORR_FWD, [l_ld, r_ld, out_ld, coff, B, H, W, Cc, md, stride, copy_left, zero_tail, 0], [], [L, R, u, out])

    def corr_fwd_prec(self, L, l_ld, R, r_ld, u, out, out_ld, coff, B, H, W, Cc, md, stride, copy_left, zero_tail, precision, stream):
        self._op(_ffi.OP_CORR_FWD, [l_ld, r_ld, out_ld, coff, B, H, W, Cc, md, stride, copy_left, zero_tail, precision], [], [L, R, u, out])

    def level_front_fwd(self, Vc, Hc, Wc, mul, L, l_ld, R, r_ld, out, out_ld, coff, Rw, rw_ld, u, B, H, W, Cc, md, zero_tail, stream):
        self._op(_ffi.OP_LEVEL_FRONT, [Hc, Wc, l_ld, r_ld, out_ld, coff, rw_ld, B, H, W, Cc, md, zero_tail, 0], [mul], [Vc, L, R, out, Rw, u])

    def level_front_fwd_planes(self, Vc, Hc, Wc, mul, L, l_ld, R, r_ld, out, out_ld, coff, Rw, rw_ld, u, B, H, W, Cc, md, zero_tail, out_hi, out_lo, out_pld, stream):
        self._op(_ffi.OP_LEVEL_FRONT, [Hc, Wc, l_ld, r_ld, out_ld, coff, rw_ld, B, H, W, Cc, md, zero_tail, out_pld], [mul], [Vc, L, R, out, Rw, u, out_hi, out_lo])

    def conv_image_fwd(self, frames, NB, H0, W0, Cc, Hp, Wp, rpt, rpl, div, sub, w, bias, N, stride, pad_t, pad_l, alpha, out, out_ld, shadow, shadow_ld, stream):
        self._op(_ffi.OP_CONV_IMAGE, [NB, H0, W0, Cc, Hp, Wp, rpt, rpl, N, stride, pad_t, pad_l, out_ld, shadow_ld], [div, sub, alpha], [frames, w, bias, out, shadow])

    def level_front_head_fwd(self, X, x_ld, K, hw, hb, Vc, Hc, Wc, mul, L, l_ld, R, r_ld, out, out_ld, coff, Rw, rw_ld, u, B, H, W, Cc, md, zero_tail, out_hi, out_lo, out_pld,
                             stream):
        self._op(_ffi.OP_LEVEL_FRONT, [Hc, Wc, l_ld, r_ld, out_ld, coff, rw_ld, B, H, W, Cc, md, zero_tail, out_pld, x_ld, K], [mul],
                 [Vc, L, R, out, Rw, u, out_hi, out_lo, X, hw, hb])

    def corr_bwd(self, g, g_ld, coff, L, l_ld, R, r_ld, dL, dl_ld, acc_l, dR, dr_ld, acc_r, du, acc_u,
                 B, H, W, Cc, md, stride, copy_left, stream):
        self.corr_bwd_prec(g, g_ld, coff, L, l_ld, R, r_ld, dL, dl_ld, acc_l, dR, dr_ld, acc_r, du, acc_u, B, H, W, Cc, md, stride, copy_left, 0, stream)

    def corr_bwd_prec(self, g, g_ld, coff, L, l_ld, R, r_ld, dL, dl_ld, acc_l, dR, dr_ld, acc_r, du, acc_u,
                      B, H, W, Cc, md, stride, copy_left, precision, stream):
        self._op(_ffi.OP_CORR_BWD, [g_ld, coff, l_ld, r_ld, dl_ld, acc_l, dr_ld, acc_r, acc_u, B, H, W, Cc, md, stride, copy_left, precision],
                 [], [g, L, R, dL, dR, du])

    def corr_warp_bwd(self, g, g_ld, coff, L, l_ld, Rw, rw_ld, img, img_ld, u, dL, dl_ld, acc_l, dimg, dimg_ld, du, B, H, W, Cc, md, stride, copy_left, stream):
        self._op(_ffi.OP_CORR_WARP_BWD, [g_ld, coff, l_ld, rw_ld, img_ld, dl_ld, acc_l, dimg_ld, B, H, W, Cc, md, stride, copy_left], [],
                 [g, L, Rw, img, u, dL, dimg, du])

    def warp_fwd(self, img, img_ld, u, out, out_ld, B, H, W, Cc, stream):
        self._op(_ffi.OP_WARP_FWD, [img_ld, out_ld, B, H, W, Cc], [], [img, u, out])

    def warp_bwd(self, g, g_ld, img, img_ld, u, dimg, dimg_ld, du, acc_u, B, H, W, Cc, stream):
        self._op(_ffi.OP_WARP_BWD, [g_ld, img_ld, dimg_ld, acc_u, B, H, W, Cc], [], [g, img, u, dimg, du])

    def resize_fwd(self, inp, out, B, Hi, Wi, Hr, Wr, cy, cx, Ho, Wo, mul, mode, stream):
        self._op(_ffi.OP_RESIZE_FWD, [B, Hi, Wi, Hr, Wr, cy, cx, Ho, Wo, mode, 0], [mul], [inp, out])

    def resize_bwd(self, g, inp, din, accumulate, B, Hi, Wi, Hr, Wr, cy, cx, Ho, Wo, mul, mode, stream):
        self._op(_ffi.OP_RESIZE_BWD, [B, Hi, Wi, Hr, Wr, cy, cx, Ho, Wo, mode, accumulate], [mul], [g, inp, din])

    def resize_image_fwd(self, inp, out, B, Hi, Wi, Cc, Ho, Wo, stream):
        self._op(_ffi.OP_RESIZE_IMAGE, [B, Hi, Wi, Cc, Ho, Wo], [], [inp, out])

    def pad_reflect(self, inp, out, B, H, W, Cc, Hp, Wp, pt, pl, out_ld, div, sub, stream):
        self._op(_ffi.OP_PAD_REFLECT, [B, H, W, Cc, Hp, Wp, pt, pl, out_ld], [div, sub], [inp, out])

    def reprojection_loss(self, left, right, disp, ws, result, ddisp, grad_scale, B, H, W, stream):
        self._op(_ffi.OP_LOSS, [B, H, W, 0], [grad_scale], [left, right, disp, ws, result, ddisp])

    def reprojection_loss_phase(self, left, right, disp, ws, result, ddisp, grad_scale, B, H, W, phase, stream):
        self._op(_ffi.OP_LOSS, [B, H, W, phase], [grad_scale], [left, right, disp, ws, result, ddisp])

    def proxy_loss(self, pred, proxy, ws, result, dpred, weight, grad_scale, B, H, W, stream):
        self._op(_ffi.OP_PROXY_LOSS, [B, H, W], [weight, grad_scale], [pred, proxy, ws, result, dpred])

    def supervised_loss(self, pred, target, ws, result, dpred, weight, grad_scale, max_disp, B, H, W, stream):
        self._op(_ffi.OP_SUPERVISED_LOSS, [B, H, W], [weight, grad_scale, max_disp], [pred, target, ws, result, dpred])

    def adam(self, var, m, v, grad, n, state, lr, beta1, beta2, eps, gs, stream):
        import struct
        self._op(_ffi.OP_ADAM, [struct.unpack("<i", struct.pack("<f", gs))[0]], [lr, beta1, beta2, eps], [var, m, v, grad, state], n=n)

    def adam_advance(self, state, beta1, beta2, stream):
        self._op(_ffi.OP_ADAM_ADVANCE, [], [beta1, beta2], [state])

    def metrics(self, disp, gt, ws, result, th, B, H, W, stream):
        self._op(_ffi.OP_METRICS, [B, H, W], [th], [disp, gt, ws, result])

    def momentum(self, var, accum, grad, n, lr, mom, gs, stream):
        self._op(_ffi.OP_MOMENTUM, [], [lr, mom, gs], [var, accum, grad], n=n)

    def copy_channels(self, src, src_ld, dst, dst_ld, npix, nch, scale, accumulate, stream):
        self._op(_ffi.OP_COPY_CH, [src_ld, dst_ld, nch, accumulate], [scale], [src, dst], n=npix)

    def leaky_bwd(self, dy, dy_ld, y, y_ld, npix, nch, alpha, stream):
        self._op(_ffi.OP_LEAKY_BWD, [dy_ld, y_ld, nch], [alpha], [dy, y], n=npix)

    def bias_grad(self, dz, dz_ld, npix, nch, db, stream):
        self._op(_ffi.OP_BIAS_GRAD, [dz_ld, nch], [], [dz, db], n=npix)

    def allreduce_sum(self, bufs, counts, n, comm, stream):
        """bufs / counts: ctypes arrays as for _ffi.Lib.allreduce_sum (the recorder copies the n pointers and counts into the op)"""
        assert 1 <= n <= 8
        self._op(_ffi.OP_ALLREDUCE, [n] + [int(counts[k]) for k in range(n)], [], [comm] + [bufs[k] for k in range(n)])

    def fetch_inputs(self, table, dst, counts, n, stream):
        """table: device-visible address of an mh_input_table; dst / counts: ctypes arrays as for _ffi.Lib.fetch_inputs"""
        assert 1 <= n <= _ffi.FETCH_MAX
        self._op(_ffi.OP_FETCH_INPUTS, [n] + [int(counts[k]) for k in range(n)], [], [table] + [dst[k] for k in range(n)])

    def bias_grad_partial(self, dz, dz_ld, npix, nch, ws, nblocks, stream):
        self._op(_ffi.OP_BIAS_GRAD, [dz_ld, nch, nblocks], [], [dz, ws], n=npix)

    def fill(self, p, n, v, stream):
        self._op(_ffi.OP_FILL, [], [v], [p], n=n)

    # -- finalise ---------------------------------------------------------------------------
    def compile(self):
        arr = (_ffi.Op * len(self.ops))(*self.ops)
        p = Plan(arr, len(self.ops), self.keep, dict(self.stats))
        p.work = dict(self.work)
        p.elided = list(getattr(self, "elided", ()))       # [(pointer, bytes)] of fp32 buffers this plan no longer writes (engine._note_elided)
        return p

    def compile_parts(self):
        """One Plan per section between cut() marks (the work statistics stay with the first)."""
        bounds = [0] + [c for c in self.cuts if 0 < c < len(self.ops)] + [len(self.ops)]
        parts = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            arr = (_ffi.Op * (b - a))(*self.ops[a:b])
            p = Plan(arr, b - a, self.keep, dict(self.stats) if not parts else {})
            p.work = {k - a: v for k, v in self.work.items() if a <= k < b}
            p.elided = list(getattr(self, "elided", ()))   # the elision is a property of the recorded step, whichever part a reader holds
            parts.append(p)
        return parts


class Plan(object):
    """An immutable op array + (optionally) its captured hipGraph."""

    def __init__(self, arr, n, keep, stats=None):
        self.arr, self.n, self.keep, self.stats = arr, n, keep, stats or {}
        self.graph = None
        self.work = {}
        self.elided = []          # [(pointer, bytes)] of fp32 buffers the recorded step no longer writes (Recorder.compile / compile_parts fill it)

    def run(self, lib, stream):
        lib.plan_run(self.arr, self.n, C.c_void_p(stream))

    def capture(self, lib, stream, copies=1):
        """Capture the plan into a hipGraph on `stream` (must not be the legacy default stream).  copies > 1: that many executable graphs of the same plan, launched
        in turn -- a replay enqueued while the previous one is in flight is then never the SAME executable (experiment r6v)."""
        s = C.c_void_p(stream)
        self.graphs = []
        for _ in range(max(1, copies)):
            lib.graph_begin(s)
            try:
                lib.plan_run(self.arr, self.n, s)
            finally:
                g = C.c_void_p()
                lib.graph_end(s, C.byref(g))
            self.graphs.append(g)
        self.graph = self.graphs[0]
        self._turn = 0

    def launch(self, lib, stream):
        if self.graph is not None:
            gs = getattr(self, "graphs", None)
            if gs and len(gs) > 1:
                self._turn = (self._turn + 1) % len(gs)
                lib.graph_launch(gs[self._turn], C.c_void_p(stream))
            else:
                lib.graph_launch(self.graph, C.c_void_p(stream))
        else:
            self.run(lib, stream)


class MultiPlan(object):
    """Several independent plans (one per private-model stream of a GPU) replayed as parallel branches of ONE hipGraph (mh_plans_run): the
    latency-bound step chains of S models share the chip instead of queueing behind one another."""

    def __init__(self, plans):
        self.plans = list(plans)
        self.refs = (_ffi.PlanRef * len(self.plans))()
        for i, p in enumerate(self.plans):
            self.refs[i].ops, self.refs[i].nops = C.addressof(p.arr), p.n
        self.graph = None
        self.n = sum(p.n for p in self.plans)

    def run(self, lib, stream):
        lib.plans_prepare(len(self.plans))
        lib.plans_run(self.refs, len(self.plans), C.c_void_p(stream))

    def capture(self, lib, stream):
        lib.plans_prepare(len(self.plans))
        s = C.c_void_p(stream)
        lib.graph_begin(s)
        try:
            lib.plans_run(self.refs, len(self.plans), s)
        finally:
            g = C.c_void_p()
            lib.graph_end(s, C.byref(g))
        self.graph = g

    def launch(self, lib, stream):
        if self.graph is not None:
            lib.graph_launch(self.graph, C.c_void_p(stream))
        else:
            self.run(lib, stream)
