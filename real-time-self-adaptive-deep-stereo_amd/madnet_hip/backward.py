"""The backward pass of MADNet as a recorded plan (Stereo_Online_Adaptation.py:85-128: gradients of the loss w.r.t. the selected variables through
Nets/MadNet.py:251-364) -- a mixin of MadNetEngine.  Order: loss head(s) -> context network -> levels 2 .. 6 (estimator, correlation + warp gradient,
the coordinate gradient into the next level's head) -> the two pyramid towers as batch 2B; filter gradients leave in batches for a side lane
(Schedule: TAIL_MAIN, PYR_FLUSH_AFTER, EARLY_WGS ...)."""
from . import ops
from .netdef import PYR, EST, CTX, LEVELS, FEAT, ALPHA, pyr_name, est_name, ctx_name, _merge_ranges      # noqa: F401


def ops_fill(lib, t, off, count):
    """record/launch a zero fill of t.flatten()[off:off+count]."""
    import ctypes as C
    flat = t.reshape(-1)
    lib.fill(C.c_void_p(flat.data_ptr() + 4 * off), count, 0.0, None)


class BackwardRecorder(object):
    def _train_flags(self, train_vars, bulkhead):
        tv = set(train_vars)
        pyr_tr = {i: (pyr_name(i) + "/weights") in tv for i in range(1, 13)}
        pyr_need = {}                       # gradient w.r.t. F_i needed?
        acc = False
        for i in range(1, 13):
            acc = acc or pyr_tr[i]
            pyr_need[i] = acc
        est_tr = {k: [(est_name(k, j) + "/weights") in tv for j in range(1, 7)] for k in LEVELS}
        ctx_tr = [(ctx_name(j) + "/weights") in tv for j in range(1, 8)]
        # anything trainable upstream of V_k (deeper levels chain only through u when not bulkhead)
        up_V = {}
        prev = False
        for k in LEVELS:
            need_u = (not bulkhead) and prev and k != 6
            up_V[k] = any(est_tr[k]) or pyr_need[FEAT[k]] or need_u
            prev = up_V[k]
        return pyr_tr, pyr_need, est_tr, ctx_tr, up_V

    def record_backward(self, r, head, train_vars, bulkhead, heads=None, early_update=None, at_cut=None):
        """at_cut(r): called where the estimators' / context network's gradients (and the loss) are final and the pyramid's backward pass starts -- the point
        build_plan(part='grad_split') cuts the recording at; the in-graph shared-model step records its first all-reduce there (engine._build_plan).
        early_update = (lr, momentum, grad_scale) (EARLY_UPDATE, FULL momentum steps): the update of a batch's layers follows the batch's reduction on its
        lane -- their input gradients were launched before the batch's fork edge and nothing later in the step reads those weights (the fragment banks
        were packed at the start of the step) -- instead of ONE launch over every parameter behind the join; returns the ranges updated that way.
        head: 'final' (loss on rescaled_prediction, FULL mode) or a level k in LEVELS
        (loss on the _make_disp of that level / of the context output for k=2, MAD mode).
        Assumes the matching d(loss)/d(disparity map) is already in self.dpred / self.ddisp_k.
        heads (offline training, Train.py:100): {'final' | level: gradient buffer} -- a loss on EVERY prediction at once;
        the per-head gradients accumulate where the heads meet (dfinal, dV[k]).
        Emits: zero of the touched gradient ranges, all needed dgrad/wgrad kernels."""
        lib, B = r, self.B
        P = self.params
        pyr_tr, pyr_need, est_tr, ctx_tr, up_V = self._train_flags(train_vars, bulkhead)
        # zero of the gradient ranges (bias gradients and single-split filter gradients accumulate): with ONE filter-gradient lane it goes
        # onto that lane -- everything that touches g runs there, behind it -- and off the critical path
        g_side = self.sched.ONE_FILL and self.wgrad_lanes == 1 and hasattr(lib, "lane")
        tail_vars = [pyr_name(1) + "/weights", pyr_name(1) + "/biases"]
        tail_lane = self.sched.TAIL_LANE if (self.sched.TAIL_SPLIT and self.sched.TAIL_LANE and g_side and pyr_tr[1] and pyr_tr[2] and all(v in train_vars for v in tail_vars)) else 0
        if g_side:
            lib.lane = 1
        try:
            for o, c in P.ranges([v for v in train_vars if not (tail_lane and v in tail_vars)]):
                ops_fill(lib, P.g, o, c)
            if tail_lane:
                lib.lane = tail_lane
                for o, c in P.ranges(tail_vars):
                    ops_fill(lib, P.g, o, c)
        finally:
            if g_side:
                lib.lane = 0
        # ONE fill for the feature gradients of all cost-volume levels (13.9 MB at 1242x375) instead of a fill in front of every level's
        # warp-gradient scatter: both towers start from zero, every contribution accumulates
        # (round 6) ... and NO fill where the fused back end is the FIRST writer of both halves: the row-owned gather kernel of mh_corr_warp_bwd stores
        # every element of the left half (dL) and of the right half (the gathered warp taps) of its level's feature gradient
        first_writer = self.sched.FIRST_WRITER and self.sched.FUSE_BACK and self.warping
        prezero = self.sched.ONE_FILL and self.warping and not bulkhead and not first_writer
        if prezero:
            ops_fill(lib, self.dF_levels, 0, self.dF_levels.numel())
        written = set()                     # gradient buffers that already hold a contribution
        if prezero:
            for i in FEAT.values():
                written.add(("F", i, 0)); written.add(("F", i, 1))
        segs = []                           # partial filter-gradient segments of this backward pass

        pending = []                        # deferred filter-gradient launches (flushed as one side-lane batch)
        batched = (self.wgrad_lanes > 0 and hasattr(lib, "lane")) or (self.use_stream and self.partial_wgrad)
        if not batched:
            early_update = None
        upd_fresh, upd_done = [], []        # early_update: parameter ranges the batch being collected completes / ranges already updated

        def wgrad(xv, dzv, base, stride=1, dil=1):
            dw, db = P.tensor(base + "/weights", "g"), P.tensor(base + "/biases", "g")
            if batched:
                pending.append((xv, dzv, dw, db, stride, dil))      # issued per batch (flush): on a side lane, and / or as one streamed launch
                if early_update is not None:
                    for t in (dw, db):
                        a = (t.data_ptr() - P.g.data_ptr()) // 4
                        assert 0 <= a and a + t.numel() <= P.total
                        upd_fresh.append((a, min((a + t.numel() + 3) & ~3, P.total)))       # (+ the tensor's alignment padding: zero gradient, zero momentum)
            elif not self.partial_wgrad:
                ops.conv2d_wgrad(lib, xv, dzv, dw, db, stride=stride, dil=dil)
            else:
                ops.conv2d_wgrad_partial(lib, self.lib, self.wsa, segs, xv, dzv, dw, db, stride=stride, dil=dil)

        nflush = [0]

        chain_stamped = [False]

        def flush(lane=None, tail=False, on_main=False):
            """Issue the deferred filter gradients as ONE batch on a side lane (one fork edge): they read only
            buffers that nothing later in the step overwrites, so they may run concurrently with everything that
            follows on lane 0 until the reduction joins them.
            tail (the flush behind the LAST input gradient, self.sched.TAIL_MAIN): nothing follows on lane 0 any more, so the batch is split -- the layers of the
            streamed kernel (conv4 .. conv2) run on lane 0 itself while the side lane does the image layer's gradient and its reduction."""
            if not pending:
                return
            # (not with early_update: the ranges a batch completes are collected per flush, not per half)
            if (tail and self.sched.TAIL_MAIN and early_update is None and self.wgrad_lanes > 0 and hasattr(lib, "lane") and self.use_stream and self.partial_wgrad
                    and ops._bwd_precision() == 1):
                streamed = [it for it in pending if ops.wgrad_stream_ok(it[0], it[1], it[2], it[4], it[5]) and it[0].npix >= self.stream_min_pix]
                rest = [it for it in pending if not any(it is q for q in streamed)]
                if streamed and rest:
                    pending[:] = rest
                    flush(lane=lane)
                    pending[:] = streamed
                    flush(on_main=True)
                    return
            side = self.wgrad_lanes > 0 and hasattr(lib, "lane") and not on_main
            if side:
                lib.lane = lane if lane else 1 + nflush[0] % self.wgrad_lanes
                lib.nodefer = nflush[0] < self.sched.NODEFER_BATCHES        # the first batches (context network, 1/4-resolution estimator) carry most of the work
            nflush[0] += 1
            try:
                self._stamp(lib, "wgrad_batch%d_start" % nflush[0])
                batch = []
                todo = list(pending)
                if self.use_stream and self.partial_wgrad and ops._bwd_precision() == 1:
                    items, casts, todo = [], [], []
                    for xv, dzv, dw, db, stride, dil in pending:
                        if ops.wgrad_stream_ok(xv, dzv, dw, stride, dil) and xv.npix >= self.stream_min_pix:
                            items.append((self._shadow(xv, casts), self._shadow(dzv, casts), dw, db, dil))
                        else:
                            todo.append((xv, dzv, dw, db, stride, dil))
                    ops.shadow_cast(lib, casts, self.dev, r.keep)
                    ops.wgrad_stream(lib, self.lib, self.wsa, batch, items, self.dev, r.keep, nwaves=(4 if self.B == 1 else 8),
                                     target_wgs=(self.sched.EARLY_WGS if (self.sched.EARLY_WGS and self.B == 1 and nflush[0] <= self.sched.EARLY_BATCHES) else None))
                for xv, dzv, dw, db, stride, dil in todo:
                    if self.partial_wgrad:
                        ops.conv2d_wgrad_partial(lib, self.lib, self.wsa, batch, xv, dzv, dw, db, stride=stride, dil=dil)
                    else:
                        ops.conv2d_wgrad(lib, xv, dzv, dw, db, stride=stride, dil=dil)
                # the batch's split reduction follows on the SAME lane: it too is off the critical path
                if batch:
                    ops.wgrad_reduce(lib, batch, self.dev, r.keep)
                if early_update is not None:
                    lr_, mom_, gs_ = early_update
                    for a, b in _merge_ranges(upd_fresh):
                        ops.momentum(lib, P.w[a:b], P.m[a:b], P.g[a:b], lr_, mom_, gs_)
                        upd_done.append((a, b))
                self._stamp(lib, "wgrad_batch%d_end" % nflush[0])
            finally:
                if side:
                    lib.lane = 0
                    lib.nodefer = False
                del pending[:]
                del upd_fresh[:]

        def acc_flag(key):
            a = key in written
            written.add(key)
            return a

        head_done = set()                   # levels whose head's input gradient went out with mh_head_bwd

        def fuse_head(k, **src):
            """dV[k] from its only source (the finer level's coordinate gradient through the x2 resize, or -- level 2 -- dfinal + the disparity
            channel of the context input's gradient) AND the input gradient of estimator k's head, in one launch; False = not applicable
            (another contribution already sits in dV[k], nothing below the head needs a gradient, switched off)."""
            if not (self.sched.FUSE_HEAD and hasattr(lib, "head_bwd") and up_V[k] and ("V", k) not in written):
                return False
            need_u_k = (not bulkhead) and k != 6 and up_V[k + 1]
            if not (any(est_tr[k][:5]) or pyr_need[FEAT[k]] or need_u_k):
                return False
            dxv, dVv = self._fv(self.dE[k][4]), self._fv(self.dV[k])
            ops.head_bwd(lib, self.W_(est_name(k, 6)), self.dV[k], dxv, mask_ref=self._fv(self.E[k][4]), mask_alpha=ALPHA,
                         accumulate_dx=acc_flag(("est", k, 5)), dV_shadow=self._out_shadow(dVv, est_name(k, 6)),
                         dx_shadow=self._out_shadow(dxv, est_name(k, 5)), **src)
            written.add(("V", k))
            head_done.add(k)
            return True

        def conv_bwd(xv, base, dzv, dxv, dx_key, x_act, stride=1, dil=1, need_dx=True, trainable=True, below=None):
            """below: the layer whose output gradient dxv is (its filter gradient reads it as dz): the input gradient's epilogue then also
            writes the bf16 shadow"""
            if trainable:
                wgrad(xv, dzv, base, stride=stride, dil=dil)
            if need_dx:
                acc = acc_flag(dx_key)
                wbt = self.banks32t.get(base) if (stride == 1 and not acc) else None
                dzs = self._fresh_shadow(dzv) if wbt is not None else None
                mks = self._fresh_shadow(x_act) if (wbt is not None and x_act is not None) else None
                if wbt is not None and dzs is not None and (x_act is None or mks is not None):
                    # one-plane walk of the planes kernel: dz from its shadow, the mask from the activation's hi plane; the result leaves as a shadow
                    # (always: the next input gradient stages it) and, until the post-pass proves that nothing reads it, as fp32
                    key = (dxv.ptr, dxv.B, dxv.H, dxv.W, dxv.C)
                    sh = self.shadows.get(key)
                    if sh is None:
                        sh = self.shadows[key] = ops.Shadow(dxv.B, dxv.H, dxv.W, dxv.C, self.dev)
                    ops.conv2d_planes_bwd(lib, dzs, self.W_(base), wbt, dx=dxv, dx_shadow=sh, mask_shadow=mks, mask_alpha=ALPHA, dil=dil)
                    self._fresh.add(key)
                    return
                ops.conv2d_dgrad(lib, dzv, self.W_(base), dxv, stride=stride, dil=dil, accumulate=acc,
                                 mask_ref=x_act, mask_alpha=ALPHA, wb=self.Wd_(base),
                                 shadow=(self._out_shadow(dxv, below) if below else None), dz_shadow=self._fresh_shadow(dzv),
                                 mask_shadow=(self._fresh_shadow(x_act) if x_act is not None else None))

        if heads is None:
            heads = {head: (self.dpred if head == "final" else self.ddisp_k)}
        start_level = 2 if ("final" in heads or 2 in heads) else min(heads)
        h2, w2, c2 = self.fshape[4]
        # ---- heads ------------------------------------------------------------------------------
        for hd in sorted(heads, key=lambda x: (0 if x == "final" else x)):
            gbuf = heads[hd]
            if hd == "final":
                ops.resize_bwd(lib, gbuf, self.final, self.dfinal, self.Hp, self.Wp, self.pt, self.pl,
                               mul=-20.0, mode=2, accumulate=acc_flag(("final",)))
            elif hd == 2:
                ops.resize_bwd(lib, gbuf, self.final, self.dfinal, self.Hp, self.Wp, self.pt, self.pl,
                               mul=-20.0, mode=1, accumulate=acc_flag(("final",)))
            else:
                ops.resize_bwd(lib, gbuf, self.V[hd], self.dV[hd], self.Hp, self.Wp, self.pt, self.pl,
                               mul=-20.0, mode=1, accumulate=acc_flag(("V", hd)))
        # ---- context network ----------------------------------------------------------------------
        if start_level == 2:
            any_below = up_V[2]
            if any(ctx_tr) or any_below:
                # final = V2 + c7 : dc7 = dfinal ; dV2 (+)= dfinal
                dz = self._fv(self.dfinal)
                for j in range(7, 0, -1):
                    xin = ops.View(self.ctx_in, B, h2, w2, c2 + 1, self.ctx_ld) if j == 1 else self._fv(self.Cx[j - 2])
                    dx = ops.View(self.dctx_in, B, h2, w2, c2 + 1, self.ctx_ld) if j == 1 else self._fv(self.dCx[j - 2])
                    need_dx = any(ctx_tr[:j - 1]) or any_below
                    conv_bwd(xin, ctx_name(j), dz, dx, ("ctx", j - 1), (None if j == 1 else self._fv(self.Cx[j - 2])),
                             dil=CTX[j - 1][1], need_dx=need_dx, trainable=ctx_tr[j - 1], below=(ctx_name(j - 1) if j > 1 else None))
                    dz = dx
                    if not need_dx:
                        break
            flush()
            if up_V[2]:
                dci = ops.View(self.dctx_in, B, h2, w2, c2 + 1, self.ctx_ld)
                if not fuse_head(2, addends=(self._fv(self.dfinal), dci.slice(c2, c2 + 1))):
                    ops.copy_channels(lib, self._fv(self.dfinal), self._fv(self.dV[2]), accumulate=acc_flag(("V", 2)))
                    ops.copy_channels(lib, dci.slice(c2, c2 + 1), self._fv(self.dV[2]), accumulate=acc_flag(("V", 2)))
                if pyr_need[4]:
                    ops.copy_channels(lib, dci.slice(0, c2), self._half(self.dF[4], False), accumulate=acc_flag(("F", 4, 0)))
        # ---- levels start_level .. 6 ---------------------------------------------------------------
        for k in LEVELS[::-1]:
            if k < start_level:
                continue
            if not up_V[k] or ("V", k) not in written:
                break
            f = FEAT[k]
            h, w, c = self.fshape[f]
            ld = self.dsi_ld[k]
            cin = c + self.D + (0 if k == 6 else 1)
            need_u = (not bulkhead) and k != 6 and up_V[k + 1]
            need_dsi = pyr_need[f] or need_u
            dz = self._fv(self.dV[k])
            for j in range(6, 0, -1):
                xin = ops.View(self.dsi[k], B, h, w, cin, ld) if j == 1 else self._fv(self.E[k][j - 2])
                dx = ops.View(self.ddsi[k], B, h, w, cin, ld) if j == 1 else self._fv(self.dE[k][j - 2])
                need_dx = any(est_tr[k][:j - 1]) or need_dsi
                if j == 6 and k in head_done:           # its input gradient is already there: only the filter gradient is left
                    if est_tr[k][5]:
                        wgrad(xin, dz, est_name(k, 6))
                    dz = dx
                    continue
                conv_bwd(xin, est_name(k, j), dz, dx, ("est", k, j - 1), (None if j == 1 else self._fv(self.E[k][j - 2])),
                         need_dx=need_dx, trainable=est_tr[k][j - 1], below=(est_name(k, j - 1) if j > 1 else None))
                dz = dx
                if not need_dx:
                    break
            if k in self.sched.EST_FLUSH_AFTER or k == 6 or not need_dsi:
                flush()
            if not need_dsi:
                break
            # correlation (+ fused concat) gradient
            Lk = self._half(self.F[f], False)
            g = ops.View(self.ddsi[k], B, h, w, ld, ld)
            dL = self._half(self.dF[f], False)
            if k == 6 or not self.warping:
                Rk = self._half(self.F[f], True)
                du = self.du[k] if (k != 6 and need_u) else None      # un-warped levels: u only feeds the estimator input
                ops.corr_bwd(lib, g, Lk, Rk, dL, self._half(self.dF[f], True), self.md, self.cstride, coff=c, du=du,
                             acc_l=acc_flag(("F", f, 0)), acc_r=acc_flag(("F", f, 1)), acc_u=False, copy_left=True)
                if du is not None:
                    s_up = 2 ** k
                    if not fuse_head(k + 1, du=self.du[k], Hr=self.Hp // s_up, Wr=self.Wp // s_up, mul=20.0 / s_up):
                        ops.resize_bwd(lib, self.du[k], self.V[k + 1], self.dV[k + 1], self.Hp // s_up, self.Wp // s_up,
                                       mul=20.0 / s_up, mode=0, accumulate=acc_flag(("V", k + 1)))
            elif self.sched.FUSE_BACK and (("F", f, 1) in written or first_writer):
                # the level's correlation + concat gradient and the warp gradient in ONE launch (mh_corr_warp_bwd): the gradient w.r.t. the warped
                # features never goes to memory; the right half of the feature gradient was zeroed by the pass's single fill / holds earlier contributions,
                # or (first_writer) is overwritten by this launch
                du = self.du[k] if need_u else None
                ops.corr_warp_bwd(lib, g, Lk, self._fv(self.Rw[k]), self._half(self.F[f], True), self.u[k], dL, self._half(self.dF[f], True), du,
                                  self.md, self.cstride, coff=c, acc_l=acc_flag(("F", f, 0)), copy_left=True, acc_img=acc_flag(("F", f, 1)))
                self._det_flush(lib, self.dF[f][B:], self.det_dF if self.deterministic else None, self.dF_levels)
                if need_u:
                    s_up = 2 ** k
                    if not fuse_head(k + 1, du=self.du[k], Hr=self.Hp // s_up, Wr=self.Wp // s_up, mul=20.0 / s_up):
                        ops.resize_bwd(lib, self.du[k], self.V[k + 1], self.dV[k + 1], self.Hp // s_up, self.Wp // s_up,
                                       mul=20.0 / s_up, mode=0, accumulate=acc_flag(("V", k + 1)))
            else:
                Rk = self._fv(self.Rw[k])
                du = self.du[k] if need_u else None
                ops.corr_bwd(lib, g, Lk, Rk, dL, self._fv(self.dRw[k]), self.md, self.cstride, coff=c, du=du,
                             acc_l=acc_flag(("F", f, 0)), acc_r=False, acc_u=False, copy_left=True)
                # warp gradient: scatter into the right tower's feature gradient (atomics -> zero first)
                dFr = self._half(self.dF[f], True)
                fresh = not acc_flag(("F", f, 1))
                if fresh:
                    ops_fill(lib, self.dF[f][B:], 0, self.dF[f][B:].numel())
                ops.warp_bwd(lib, self._fv(self.dRw[k]), self._half(self.F[f], True), self.u[k], dFr,
                             du=du, acc_u=True)
                self._det_flush(lib, self.dF[f][B:], self.det_dF if self.deterministic else None, self.dF_levels)
                if need_u:
                    # u_k = resize(V_{k+1}) * 20/2^k   (MadNet.py:274: u_{k} built at level k+1 with scales[k])
                    s_up = 2 ** k
                    if not fuse_head(k + 1, du=self.du[k], Hr=self.Hp // s_up, Wr=self.Wp // s_up, mul=20.0 / s_up):
                        ops.resize_bwd(lib, self.du[k], self.V[k + 1], self.dV[k + 1], self.Hp // s_up, self.Wp // s_up,
                                       mul=20.0 / s_up, mode=0, accumulate=acc_flag(("V", k + 1)))
        # ---- pyramid towers (batch 2B, shared weights) ---------------------------------------------
        # split point of build_plan(part='grad_split'): every gradient of the estimators / the context network is final here (their
        # batches were flushed level by level), the pyramid's come after -- the shared-model step all-reduces the first range while
        # the second is still being computed
        if hasattr(r, "cut"):
            r.cut()
        if at_cut is not None:
            at_cut(r)
        top = None
        for i in range(12, 0, -1):
            if ("F", i, 0) in written or ("F", i, 1) in written or ("Fd", i) in written:
                top = i
                break
        if top is not None and pyr_need[top]:
            # features that feed only the cost volume still need their own leaky gradient
            if ("Fd", top) not in written:
                if ("F", top, 0) not in written:
                    ops_fill(lib, self.dF[top][:B], 0, self.dF[top][:B].numel())
                if ("F", top, 1) not in written:
                    ops_fill(lib, self.dF[top][B:], 0, self.dF[top][B:].numel())
                ops.leaky_bwd(lib, self._fv(self.dF[top]), self._fv(self.F[top]), ALPHA)
            for i in range(top, 0, -1):
                if not pyr_need[i]:
                    break
                xin = ops.View(self.X0, 2 * B, self.Hp, self.Wp, 3, 4) if i == 1 else self._fv(self.F[i - 1])
                need_dx = i > 1 and pyr_need[i - 1]
                accumulate = False
                if need_dx:
                    has_l, has_r = ("F", i - 1, 0) in written, ("F", i - 1, 1) in written
                    accumulate = has_l or has_r
                    if accumulate and not has_l:
                        ops_fill(lib, self.dF[i - 1][:B], 0, self.dF[i - 1][:B].numel())
                    if accumulate and not has_r:
                        ops_fill(lib, self.dF[i - 1][B:], 0, self.dF[i - 1][B:].numel())
                if pyr_tr[i]:
                    wgrad(xin, self._fv(self.dF[i]), pyr_name(i), stride=PYR[i - 1][2])
                if (self.sched.TAIL_SPLIT and i == 2) or i in self.sched.PYR_FLUSH_BEFORE:
                    flush()                 # (TAIL_SPLIT: conv4 .. conv2 beside conv2's input gradient, not behind it)
                if need_dx:
                    # dF[i-1] is complete after this launch (the cost-volume contributions were written earlier): it is the dz of layer i - 1
                    sh = self._out_shadow(self._fv(self.dF[i - 1]), pyr_name(i - 1)) if (i - 1 > 1) else None       # (conv1's 3-channel input keeps the tiled kernel)
                    s2acc = accumulate and PYR[i - 1][2] == 2 and self.sched.PLANES_S2_ACC      # (the stride-2 parity-class kernel adds onto earlier contributions)
                    wbt = self.banks32t.get(pyr_name(i)) if (not accumulate or s2acc) else None       # (stride-2 layers: only those _bank_plan gave a bank)
                    dzs = self._fresh_shadow(self._fv(self.dF[i])) if wbt is not None else None
                    mks = self._fresh_shadow(self._fv(self.F[i - 1])) if wbt is not None else None
                    if wbt is not None and dzs is not None and mks is not None and sh is not None:
                        ops.conv2d_planes_bwd(lib, dzs, self.W_(pyr_name(i)), wbt, dx=self._fv(self.dF[i - 1]), dx_shadow=sh, mask_shadow=mks, mask_alpha=ALPHA,
                                              stride=PYR[i - 1][2], accumulate=accumulate)
                        if i in self.sched.PYR_FLUSH_AFTER:
                            flush(lane=(tail_lane if i == 1 else None))
                        continue
                    ops.conv2d_dgrad(lib, self._fv(self.dF[i]), self.W_(pyr_name(i)), self._fv(self.dF[i - 1]),
                                     stride=PYR[i - 1][2], accumulate=accumulate, mask_ref=self._fv(self.F[i - 1]),
                                     mask_alpha=ALPHA, wb=self.Wd_(pyr_name(i)), shadow=sh,
                                     dz_shadow=self._fresh_shadow(self._fv(self.dF[i])), mask_shadow=self._fresh_shadow(self._fv(self.F[i - 1])))
                if i in self.sched.PYR_FLUSH_AFTER:
                    if i == 1:
                        self._stamp(lib, "chain_end")           # lane 0: the last input gradient is behind us (the tail flush may put work on lane 0 again)
                        chain_stamped[0] = True
                    flush(lane=(tail_lane if i == 1 else None), tail=(i == 1))
        flush()
        if not chain_stamped[0]:
            self._stamp(lib, "chain_end")                       # lane 0: the last input gradient is behind us
        ops.wgrad_reduce(lib, segs, self.dev, r.keep)          # (serial variant only: the side-lane batches reduce themselves)
        r.join_next = True                                      # whatever comes next (the optimizer) waits for the side lanes
        self._stamp(lib, "joined")                              # (takes the join edge: every side lane has finished)
        r.join_next = True
        if self.deterministic:
            assert not upd_done, "deterministic mode: no early update (the bias gradients are still in their fixed-point twins)"
            self._det_flush(lib, self.params.g, self.det_g, self.params.g)
            r.join_next = True
        return _merge_ranges(upd_done)
