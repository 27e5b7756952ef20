"""MADNet executor for MI355X: static buffers + recorded forward / backward / update plans.

Host-side mirror of Nets/MadNet.py:251-364 (graph), Losses/loss_factory.py:353-395 (loss),
Stereo_Online_Adaptation.py:68-128 (loss / validation / train-op construction).  The reference
builds a TF1 graph and lets Session.run prune it to the fetches; here every (mode, block)
combination is compiled once into an op array (plan.py) and replayed natively / as a hipGraph.

Memory layout (all float32, NHWC, resident in HBM for the life of the engine):
  * one flat parameter buffer + one momentum buffer + one gradient buffer with identical
    layout; MAD blocks are contiguous ranges so the update is one fused launch per range;
  * both pyramid towers run as batch 2B (shared weights => their weight gradients sum for free);
  * the estimator input [reference | corr | upsampled disparity] is ONE buffer written by the
    correlation kernel (no tf.concat copies); channel counts are padded to multiples of 4 so all
    row accesses are 16-byte vectors.
"""
import os

import numpy as np
import torch

from . import ops
from .plan import Recorder
from .schedule import Schedule
from .elision import ElisionPasses
from .backward import BackwardRecorder, ops_fill      # noqa: F401

from .netdef import PYR, EST, CTX, LEVELS, FEAT, ALPHA, _r4, pyr_name, est_name, ctx_name, _merge_ranges, madnet_manifest      # noqa: E402,F401  (re-exported: E.LEVELS ...)








class Params(object):
    """Flat fp32 weight / momentum / gradient buffers + name -> (offset, shape) manifest."""

    def __init__(self, manifest, device):
        self.manifest = manifest
        self.offset, self.shape = {}, {}
        off = 0
        for name, shp in manifest:
            self.offset[name], self.shape[name] = off, tuple(shp)
            off += (int(np.prod(shp)) + 3) // 4 * 4          # keep every tensor 16-byte aligned
        self.total = off
        self.w = torch.zeros(off, device=device)
        self.m = torch.zeros(off, device=device)
        # + 4 floats behind the gradients: the step's loss result lives there, so the shared-model mode all-reduces the
        # gradients AND the loss that drives the reward / reset logic with ONE collective (adapter.py)
        self.g_loss = torch.zeros(off + 4, device=device)     # [gradients | loss result (4 floats)]
        self.g = self.g_loss[:off]
        self.w0 = None                                         # reset copy (restore target)

    def numel(self, name):
        return int(np.prod(self.shape[name]))

    def tensor(self, name, which="w"):
        buf = getattr(self, which)
        o = self.offset[name]
        return buf[o:o + self.numel(name)].view(self.shape[name])

    def load(self, weights):
        """weights: {name: ndarray / tensor} (HWIO); missing names keep their value."""
        for name, v in weights.items():
            if name in self.offset:
                self.tensor(name).copy_(torch.as_tensor(v, dtype=torch.float32).reshape(self.shape[name]))

    def export(self):
        return {name: self.tensor(name).detach().cpu().numpy().copy() for name, _ in self.manifest}

    def ranges(self, names):
        """Coalesced (offset, count) ranges covering the given variables."""
        spans = sorted((self.offset[n], (self.numel(n) + 3) // 4 * 4) for n in set(names))
        out = []
        for o, c in spans:
            if out and out[-1][0] + out[-1][1] == o:
                out[-1][1] += c
            else:
                out.append([o, c])
        return [(o, c) for o, c in out]


class MadNetEngine(ElisionPasses, BackwardRecorder):
    def __init__(self, lib, H, W, B=1, device="cuda", radius_d=2, stride=1, warping=True, weights=None, precision="fp32", schedule=None):
        """precision: 'fp32' = exact fp32 MFMA (parity path, default) | 'bf16' = bf16 MFMA inputs with fp32
        accumulation in the conv forward / input-gradient kernels (throughput mode; tensors stay fp32).
        schedule: a madnet_hip.schedule.Schedule (immutable; default = the committed bench line's): how this engine's plans are recorded."""
        self.sched = schedule if schedule is not None else Schedule()
        if precision not in ops.PRECISION_CODES:
            raise ValueError("precision must be one of %s" % sorted(ops.PRECISION_CODES))
        self.precision = precision
        # warping=False (MadNet.py:282-285,301-304,...): the right features enter the cost volume un-warped; the upsampled
        # disparity still joins the estimator input
        self.warping = bool(warping)
        self.lib, self.dev = lib, device
        _td = torch.device(device)
        if _td.type == "cuda" and hasattr(lib, "ensure_init"):
            with torch.cuda.device(_td):                  # the per-device set-up of the library, with THIS engine's device current (a process may drive several)
                lib.ensure_init(torch.cuda.current_device())
        self.B, self.H0, self.W0 = B, H, W
        self.md, self.cstride = radius_d, stride
        self.D = 2 * radius_d // stride + 1
        self.Hp = H if H % 64 == 0 else (H // 64 + 1) * 64          # preprocessing.pad_image(., 64)
        self.Wp = W if W % 64 == 0 else (W // 64 + 1) * 64
        self.pt, self.pl = (self.Hp - H) // 2, (self.Wp - W) // 2
        self.params = Params(madnet_manifest(radius_d, stride), device)
        if weights is not None:
            self.params.load(weights)
        self._alloc()
        self._plans = {}
        self._zeros_needed = []
        # filter gradients: atomic-free split reduction (ops.conv2d_wgrad_partial) unless switched off
        self.partial_wgrad = True            # False (tests): splits accumulate with fp32 atomics straight into g (measured slower: +10 %)
        # ... recorded on a side lane: the filter gradients are off the critical path (only the optimizer needs
        # them), so they overlap with the input-gradient chain as a parallel branch of the hipGraph
        # ONE lane since mh_plan_run defers side launches past the next lane-0 op (2.08 ms against 2.28 ms with two lanes, 2.20 ms with
        # two lanes undeferred: profiles/r02_experiments.txt #16)
        self.wgrad_lanes = int(os.environ.get("MH_WGRAD_LANES", "1"))
        # MAD plans (one block's backward pass: six to thirteen filter gradients) run as ONE serial chain unless MH_WGRAD_LANES says otherwise: the side
        # stream costs the captured graph more than the overlap of so few launches returns (round 4: 0.896 / 0.889 -> 0.877 / 0.880 ms per MAD step)
        self.mad_serial = "MH_WGRAD_LANES" not in os.environ
        # one launch per level for the inter-level upsample + warp + cost volume + concat (mh_level_front_fwd) instead of three
        self.fuse_front = True
        # split-bf16 3x3 layers of the 1/4- and 1/8-resolution estimators and the context network stream their weights from MFMA
        # fragment banks (mh_conv2d_wb), re-packed by ONE launch at the start of every step
        # ... and (bf16 / mixed) the layers of the 1/16-1/64 levels -- forward and input gradient -- take the small-layer bank kernel
        self.use_bank = precision in ("mixed", "bf16") and os.environ.get("MH_CONV_BANK", "1") != "0"
        self.bank_small_maxpix = 4096           # = the library default (mh_tune_conv_bank): the banks are packed for the layers that kernel takes
        self.bank_min_n = 32
        self.banks = {}
        self.banks_d = {}
        self.wsa = ops.WgradWorkspace(device)
        # bf16 backward: the filter gradients of the stride-1 3x3 layers run on the streaming kernel (mh_wgrad_stream: one launch per batch,
        # operands from bf16 shadows of the activations / gradient maps); MH_WGRAD_STREAM=0 keeps the tiled kernels
        self.use_stream = precision in ("mixed", "bf16") and os.environ.get("MH_WGRAD_STREAM", "1") != "0"
        self.stream_min_pix = 0
        self.shadows = {}                   # (data pointer, B, H, W, C) -> ops.Shadow, allocated once per engine
        # ... written by the epilogue of the kernel that produces the tensor (mh_conv2d_sh) wherever a conv kernel is the producer; the rest
        # (cost-volume buffers, heads, the top pyramid gradient) go through one mh_shadow_cast per batch.  fuse_shadows = False: cast everything
        self.fuse_shadows = True
        self._fresh = set()                 # shadows a producer wrote in the plan being recorded
        self._stream_train = set()          # trainable variables of that plan
        self.use_planes = self.use_bank and precision == "mixed" and self.sched.USE_PLANES
        self.banks32 = {}                   # layer -> fragment bank in the 32x32x16 image (mh_pack_weights trans = 2)
        self.banks32t = {}                  # layer -> the input gradient's one-plane bank in that image (trans = 3)
        self.planes = {}                    # (data pointer, B, H, W, C) -> ops.Planes (hi = the entry of self.shadows)
        self._fresh_planes = set()          # planes (hi AND lo) a producer wrote in the plan being recorded

    # ---------------------------------------------------------------------------------------
    def _buf(self, *shape):
        return torch.zeros(*shape, device=self.dev)

    def _alloc(self):
        B, B2 = self.B, 2 * self.B
        z = self._buf
        self.lr = z(2 * B, self.H0, self.W0, 3)                    # both frames in one buffer: ONE padding launch for the pair
        self.left, self.right = self.lr[:B], self.lr[B:]
        self.gt = z(B, self.H0, self.W0)
        self.X0 = z(B2, self.Hp, self.Wp, 4)
        self.F, self.dF = {}, {}
        h, w = self.Hp, self.Wp
        self.fshape = {}
        for i, (ci, co, s) in enumerate(PYR, 1):
            h, _, _ = ops.same_pad(h, 3, s); w, _, _ = ops.same_pad(w, 3, s)
            self.fshape[i] = (h, w, co)
            self.F[i] = z(B2, h, w, co)
        # gradients of the features: those of the five cost-volume levels live in ONE flat buffer so that a backward pass zeroes all of
        # them (the right towers' warp-gradient scatter targets, halves no head reaches in MAD mode) with one launch instead of one per level
        lv = sorted(FEAT.values())
        tot = sum(B2 * self.fshape[i][0] * self.fshape[i][1] * self.fshape[i][2] for i in lv)
        self.dF_levels = z(tot)
        off = 0
        for i in range(1, 13):
            hh, ww, co = self.fshape[i]
            if i in lv:
                n = B2 * hh * ww * co
                self.dF[i] = self.dF_levels[off:off + n].view(B2, hh, ww, co)
                off += n
            else:
                self.dF[i] = z(B2, hh, ww, co)
        self.deterministic = self.sched.DETERMINISTIC
        ops.check_planes_rule(self.lib)
        self._det_bases = []
        if self.deterministic:
            self.det_g = torch.zeros(self.params.total, dtype=torch.int64, device=self.dev)
            self.det_dF = torch.zeros(self.dF_levels.numel(), dtype=torch.int64, device=self.dev)
            import ctypes as _C
            for base, n, twin in ((self.params.g, self.params.total, self.det_g), (self.dF_levels, self.dF_levels.numel(), self.det_dF)):
                self.lib.deterministic_add(_C.c_void_p(base.data_ptr()), n, _C.c_void_p(twin.data_ptr()))
                self._det_bases.append(base.data_ptr())
        self.Rw, self.dRw, self.dsi, self.ddsi, self.E, self.dE, self.V, self.dV, self.u, self.du = ({} for _ in range(10))
        self.dsi_ld = {}
        for k in LEVELS:
            h, w, c = self.fshape[FEAT[k]]
            ld = _r4(c + self.D + (0 if k == 6 else 1))
            self.dsi_ld[k] = ld
            self.dsi[k] = z(B, h, w, ld); self.ddsi[k] = z(B, h, w, ld)
            if k != 6:
                self.Rw[k] = z(B, h, w, c); self.dRw[k] = z(B, h, w, c)
                self.u[k] = z(B, h, w); self.du[k] = z(B, h, w)
            self.E[k] = [z(B, h, w, co) for co in EST[:-1]]
            self.dE[k] = [z(B, h, w, co) for co in EST[:-1]]
            self.V[k] = z(B, h, w); self.dV[k] = z(B, h, w)
        h, w, c = self.fshape[4]
        self.ctx_ld = _r4(c + 1)
        self.ctx_in = z(B, h, w, self.ctx_ld); self.dctx_in = z(B, h, w, self.ctx_ld)
        self.Cx = [z(B, h, w, co) for co, _ in CTX[:-1]]
        self.dCx = [z(B, h, w, co) for co, _ in CTX[:-1]]
        self.final = z(B, h, w); self.dfinal = z(B, h, w)
        self.pred = z(B, self.H0, self.W0); self.dpred = z(B, self.H0, self.W0)
        self.disp_k = {k: z(B, self.H0, self.W0) for k in LEVELS}      # _make_disp outputs (k=2: context)
        self.ddisp_k = z(B, self.H0, self.W0)
        self.loss_ws = z(self.lib.loss_ws_floats(B, self.H0, self.W0))
        self.loss_ws_k = z(self.lib.loss_ws_floats(B, self.H0, self.W0))
        self.met_ws = z(self.lib.metrics_ws_floats(B, self.H0, self.W0))
        self.res_loss = self.params.g_loss[self.params.total:self.params.total + 4]
        self.res_loss_k = z(4); self.res_met = z(4)
        # continual-adaptation variant (loss_kind = 'proxy'): proxy labels + the mean_l1 loss workspace
        self.proxy = z(B, self.H0, self.W0)
        self.proxy_ws = z(self.lib.proxy_ws_floats(B, self.H0, self.W0))
        self.loss_kind = "reprojection"
        self.rscale = 1

    def set_reprojection_scale(self, s):
        """--reprojectionScale s (Stereo_Online_Adaptation.py:22-23,91-95): the MAD blocks' losses are computed on the frames
        resized to (H//s, W//s) and the block's prediction resized to the same size -- with its VALUES unchanged (the
        multiplier at :102 is H_left // H_p = 1 for the full-resolution predictions and the factor inside the loss is
        W_left_s / W_p_s = 1): replicated as written."""
        s = int(s)
        if s < 1:
            raise ValueError("reprojectionScale must be >= 1")
        self.rscale = s
        if s != 1:
            B, Hs, Ws = self.B, self.H0 // s, self.W0 // s
            z = self._buf
            self.left_s = z(B, Hs, Ws, 3); self.right_s = z(B, Hs, Ws, 3)
            self.p_s = z(B, Hs, Ws); self.dp_s = z(B, Hs, Ws)
            self.loss_ws_s = z(self.lib.loss_ws_floats(B, Hs, Ws))
        self._plans = {}

    # views -----------------------------------------------------------------------------------
    def _fv(self, t):
        return ops.view(t)

    def _half(self, t, right):
        """left (first B) or right (last B) tower half of a batch-2B pyramid tensor."""
        B = self.B
        sub = t[B:] if right else t[:B]
        return ops.view(sub)

    def W_(self, base):
        return self.params.tensor(base + "/weights")

    def b_(self, base):
        return self.params.tensor(base + "/biases")

    def Wb_(self, base):
        """MFMA fragment bank of the layer for the forward pass (None: the layer does not run a bank kernel)"""
        return self.banks.get(base)

    def Wd_(self, base):
        """... for the input gradient (small layers, bf16 backward)"""
        return self.banks_d.get(base)

    def _pyr_code(self, i, fcode=None):
        """forward precision code of pyramid layer i: in 'mixed' conv7 .. conv12 (1/16 resolution and below) run plain bf16 -- rounding ONE of them
        to bf16 moves the final disparity by 7e-5 (conv7), 6.8e-5 (conv8), 8e-6 (conv9 .. conv12) px, 1.7e-4 px together (per-layer map,
        profiles/r02_precision_map.txt); conv1 .. conv6 (9e-3 .. 9e-4 each) keep split-bf16 / exact fp32.  None = the mode's code."""
        if self.precision == "mixed" and i >= self.sched.PYR_BF16_FROM:
            return 1
        return fcode

    def _bank_plan(self):
        """[(layer, planes, trans)]: which fragment banks this engine packs every step.  Forward: planes follow the precision code the
        layer runs (2 = split-bf16 -> the 64x128 / 128x64 bank kernel or, <= bank_small_maxpix output pixels, the small-layer kernel;
        1 = bf16 -> small-layer kernel only).  Input gradient (trans 1, bf16): small layers."""
        if not self.use_bank:
            return []
        fcode, bcode = ops.PRECISION_CODES[self.precision]
        shapes = dict(self.params.manifest)
        B = self.B
        layers = []          # (name, forward precision code, output pixels)
        stride2 = set()
        s2_planes = []
        for i in range(2, 13):                                      # the stride-1 pyramid layers and the small stride-2 ones (conv7 / 9 / 11)
            h, w = (self.fshape[i][0], self.fshape[i][1])
            if PYR[i - 1][2] == 2:
                if 2 * B * h * w > self.bank_small_maxpix:
                    # (round 6) the LARGE stride-2 layers named by Schedule.PLANES_S2_FWD run the stride-2 plane kernel (split-bf16 from their producer's planes)
                    # instead of the exact-fp32 tiled kernel: a 32x32x16 bank, forward only
                    if (i in self.sched.PLANES_S2_FWD and self.use_planes and fcode == 2 and self._pyr_code(i, fcode) == 2
                            and ops.conv2d_planes_ok(self.lib, self._fv(self.F[i - 1]), self.W_(pyr_name(i)), 1, stride=2)):
                        s2_planes.append(pyr_name(i))
                    continue
                stride2.add(pyr_name(i))
            layers.append((pyr_name(i), self._pyr_code(i, fcode), 2 * B * h * w))
        for k in LEVELS:
            h, w, _ = self.fshape[FEAT[k]]
            code = 1 if (self.precision == "mixed" and k >= 4) else fcode
            layers += [(est_name(k, j), code, B * h * w) for j in range(1, 7)]
        h, w, _ = self.fshape[4]
        layers += [(ctx_name(j), fcode, B * h * w) for j in range(1, 8)]
        plan = [(n, 2, 2) for n in s2_planes]
        for n, code, pix in layers:
            kh, _, K, N = shapes[n + "/weights"]
            if kh != 3:
                continue
            small = pix <= self.bank_small_maxpix and 9 * ((K + 31) // 32) <= 64
            if code == 2 and not small and n not in stride2 and self._planes_layer(K, N):
                plan.append((n, 2, 2))
            elif code == 2 and N >= 16 and K >= 16 and (small or (N >= self.bank_min_n and K >= self.bank_min_n)):
                plan.append((n, 2, 0))
            elif code == 1 and small and N >= 16 and K >= 16:
                plan.append((n, 1, 0))
            if n in stride2:
                continue                                            # (forward only: the stride-2 input gradient runs parity classes on the tiled kernel)
            if bcode == 1 and pix > self.bank_small_maxpix and PYR[int(n.rsplit("conv", 1)[1]) - 1][2] == 1 if "pyramid" in n else (bcode == 1 and pix > self.bank_small_maxpix):
                if self._planes_bwd_layer(K, N):
                    plan.append((n, 1, 3))
                    continue
            if bcode == 1 and pix <= 2 * self.bank_small_maxpix and K >= 16 and N >= 16 and 9 * ((N + 31) // 32) <= 64:
                plan.append((n, 1, 1))
        # the LARGE stride-2 pyramid layers whose input gradient is the first contribution to its target (conv3: F2 feeds no cost volume): the parity-class
        # plane kernel (mh_conv2d_planes_bwd on a stride-2 descriptor) from the one-plane mirrored / transposed bank
        if bcode == 1 and self.use_planes and self.sched.PLANES_DGRAD:
            for i in range(3, 13):
                h, w = self.fshape[i][0], self.fshape[i][1]
                # (round 6: the parity-class kernel accumulates, so a layer whose input is a cost-volume level -- conv5: F4 already holds the correlation's gradient -- qualifies)
                if PYR[i - 1][2] == 2 and 2 * B * h * w > self.bank_small_maxpix and ((i - 1) not in FEAT.values() or self.sched.PLANES_S2_ACC):
                    _, _, K, N = shapes[pyr_name(i) + "/weights"]
                    dxv = self._fv(self.dF[i - 1])
                    if ops.conv2d_planes_bwd_ok(self.lib, dxv, self.W_(pyr_name(i)), 1, stride=2):
                        plan.append((pyr_name(i), 1, 3))
        return plan

    def _stamp(self, lib, label):
        if not self.sched.STAMPS or not hasattr(lib, "stamp"):
            return
        if getattr(self, "stamps", None) is None:
            self.stamps = torch.zeros(64, dtype=torch.int64, device=self.dev)
        self.stamp_labels.append((label, getattr(lib, "lane", 0)))
        ops.stamp(lib, self.stamps, len(self.stamp_labels) - 1)

    def _planes_layer(self, K, N):
        """does mh_conv2d_planes have an instance for a stride-1 3x3 layer with K input / N output channels?  (csrc/conv_planes.hip)"""
        return self.use_planes and N % 8 == 0 and N <= 128 and ((K + 15) // 16) in (2, 3, 4, 5, 6, 8)

    def _planes_bwd_layer(self, K, N):
        """does mh_conv2d_planes_bwd have an instance for the input gradient of a stride-1 3x3 layer K -> N?"""
        if not (self.use_planes and self.sched.PLANES_DGRAD):
            return False
        import ctypes as C
        d = ops.conv_desc(1, 8, 8, 8, 8, K, N, 3, 3, 1, 1, 1, 1, 0, 0, K, 0, precision=1)
        return self.lib.conv2d_planes_bwd_ok(C.byref(d)) == 1

    def _bank_of(self, trans):
        return {0: self.banks, 1: self.banks_d, 2: self.banks32, 3: self.banks32t}[trans]

    def _planes_of(self, v):
        """the Planes object of View v (allocated on first use; its hi plane is v's Shadow)"""
        key = (v.ptr, v.B, v.H, v.W, v.C)
        pl = self.planes.get(key)
        if pl is None:
            sh = self.shadows.get(key)
            if sh is None:
                sh = self.shadows[key] = ops.Shadow(v.B, v.H, v.W, v.C, self.dev)
            pl = self.planes[key] = ops.Planes(sh, self.dev)
        return key, pl

    def _in_planes(self, lib, v, r):
        """planes of an input View: as a producer of this plan left them, else split here (one launch; tensors no plane-writing kernel produces)"""
        key, pl = self._planes_of(v)
        if key not in self._fresh_planes:
            ops.plane_split(lib, [(v, pl)], self.dev, r.keep)
            self._fresh_planes.add(key)
            self._fresh.add(key)                # the hi plane is the tensor's bf16 shadow: no cast in the backward pass
        return pl

    def _conv_fwd(self, lib, r, x, base, o, stride=1, dil=1, alpha=ALPHA, precision=None, shadow_consumer=None):
        """forward conv of layer `base`: from planes (mh_conv2d_planes) where the layer has a 32x32x16 bank, else the fp32-operand kernels"""
        wb32 = self.banks32.get(base)              # (stride 2: only the layers _bank_plan gave a 32x32x16 bank -- Schedule.PLANES_S2_FWD)
        if wb32 is not None:
            xp = self._in_planes(lib, x, r)
            key, op_ = self._planes_of(o)
            ops.conv2d_planes(lib, xp, self.W_(base), wb32, self.b_(base), out=o, out_planes=op_, dil=dil, alpha=alpha, stride=stride)
            self._fresh_planes.add(key)
            self._fresh.add(key)
            return
        if shadow_consumer and shadow_consumer in self.banks32 and self.sched.FUSE_SPLITS:
            # the consumer runs from planes: this layer's epilogue writes them (mh_conv2d_sh4) instead of a split launch in front of the consumer
            key, op_ = self._planes_of(o)
            ops.conv2d_fwd(lib, x, self.W_(base), self.b_(base), o, stride=stride, dil=dil, alpha=alpha, wb=self.Wb_(base), precision=precision,
                           out_planes=op_)
            self._fresh_planes.add(key)
            self._fresh.add(key)
            return
        sh = self._out_shadow(o, shadow_consumer) if shadow_consumer else None
        ops.conv2d_fwd(lib, x, self.W_(base), self.b_(base), o, stride=stride, dil=dil, alpha=alpha,
                       wb=self.Wb_(base), precision=precision, shadow=sh)

    def record_forward(self, r, make_disps=(), need_x0=True):
        """need_x0 = False: nothing of the plan reads the padded frames X0 after the forward pass (no backward pass): with Schedule.IMAGE_CONV they are not written"""
        B, lib = self.B, r
        self._x0_pending = False
        head2_fused = False
        pending_head = None
        self._fresh_planes = set()
        self._stamp(lib, "start")
        if self.use_bank:
            plan = self._bank_plan()
            for n, planes, trans in plan:
                tgt = self._bank_of(trans)
                if n not in tgt:
                    tgt[n] = torch.zeros(ops.pack_bytes(self.W_(n), planes, trans) // 4, device=self.dev)
            # (in line: on a side lane beside the first pyramid layers, which read no bank, it measured no gain -- profiles/r03_experiments.txt; PACK_SIDE
            #  repeats that experiment: the launch on lane 1 beside pad_reflect + conv1 (28 us), joined in front of conv2)
            side_pack = self.sched.PACK_SIDE and hasattr(lib, "lane") and self.wgrad_lanes > 0
            if side_pack:
                lib.lane = 1
            try:
                ops.pack_weights(lib, [(self.W_(n), self._bank_of(trans)[n], planes, trans) for n, planes, trans in plan],
                                 self.dev, r.keep)
            finally:
                if side_pack:
                    lib.lane = 0
        image_conv = (self.sched.IMAGE_CONV and hasattr(self.lib, "conv_image_ok") and self.lib.conv_image_ok(PYR[0][0], PYR[0][1], 3, 3, PYR[0][2]) == 1)
        if image_conv:
            # the padded copy: only conv1's filter gradient reads it (need_x0: the plan has a backward pass) -- off the chain, on lane 1 when there is one
            if need_x0:
                # ONE filter-gradient lane + the side reductions of the loss: the padding launch rides on lane 1 in front of those (record_loss_metrics) -- no fork
                # edge of its own (a fork at the head of the step cost lane 0 most of what the launch gave back: r05_experiments.txt #15), and conv1's filter
                # gradient runs on that lane, behind it.  Any other lane layout: in line, as before.
                self._x0_pending = bool(hasattr(lib, "lane") and self.wgrad_lanes == 1 and self.sched.SIDE_LOSS and self.loss_kind != "proxy" and not self.sched.TAIL_SPLIT)
                if not self._x0_pending:
                    ops.pad_reflect(lib, self.lr, self.X0, self.pt, self.pl)
        else:
            ops.pad_reflect(lib, self.lr, self.X0, self.pt, self.pl)
        x = ops.View(self.X0, 2 * B, self.Hp, self.Wp, 3, 4)
        for i, (ci, co, s) in enumerate(PYR, 1):
            o = self._fv(self.F[i])
            if i == 1 and image_conv:
                ops.conv_image_fwd(lib, self.lr, self.Hp, self.Wp, self.pt, self.pl, self.W_(pyr_name(1)), self.b_(pyr_name(1)), o, stride=s, alpha=ALPHA,
                                   shadow=self._out_shadow(o, pyr_name(2)))
                x = o
                continue
            if i == 2 and self.use_bank and self.sched.PACK_SIDE and hasattr(lib, "lane") and self.wgrad_lanes > 0:
                lib.join_lanes_next = 1 << 1                  # conv1 (3 input channels) never has a bank: every later layer waits for the packing
            # F_i is the input of layer i + 1 (stride 1 or 2: both streamed)
            self._conv_fwd(lib, r, x, pyr_name(i), o, stride=s, precision=self._pyr_code(i), shadow_consumer=(pyr_name(i + 1) if i < 12 else None))
            x = o
        for k in LEVELS:
            f = FEAT[k]
            h, w, c = self.fshape[f]
            Lk = self._half(self.F[f], False)
            Rk = self._half(self.F[f], True)
            ld = self.dsi_ld[k]
            dsi = ops.View(self.dsi[k], B, h, w, ld, ld)
            fused = k != 6 and self._front_fused()
            head = pending_head
            pending_head = None
            if fused:
                # u_k = resize(V_{k+1}) * 20 / 2^k (MadNet.py:274), warp, cost volume + concat: one launch
                xin = ops.View(self.dsi[k], B, h, w, c + self.D + 1, ld)
                pl = None
                if self.sched.FUSE_SPLITS and self.use_planes and self.cstride == 1:
                    key, pl_ = self._planes_of(xin)
                    if est_name(k, 1) in self.banks32:
                        pl = pl_                                  # hi + lo: the estimator's first layer runs from planes
                        self._fresh_planes.add(key); self._fresh.add(key)
                    elif (est_name(k, 1) + "/weights") in self._stream_train and self.use_stream and self.partial_wgrad and ops._bwd_precision() == 1:
                        pl = pl_.hi                               # hi only: the shadow its streamed filter gradient reads (no cast in the backward pass)
                        self._fresh.add(key)
                if head is not None:
                    # ... and the disparity head of level k + 1 (recorded nowhere else: see below)
                    hx, hname = head
                    ops.level_front_head_fwd(lib, hx, self.W_(hname), self.b_(hname), self.V[k + 1], 20.0 / 2 ** k, Lk, Rk, dsi, self._fv(self.Rw[k]), self.u[k], self.md,
                                             coff=c, planes=pl)
                    if (k + 1) in make_disps:
                        self._make_disp(lib, self.V[k + 1], self.disp_k[k + 1])
                else:
                    ops.level_front_fwd(lib, self.V[k + 1], 20.0 / 2 ** k, Lk, Rk, dsi, self._fv(self.Rw[k]), self.u[k], self.md, coff=c, planes=pl)
            else:
                if k != 6 and self.warping:
                    ops.warp_fwd(lib, Rk, self.u[k], self._fv(self.Rw[k]))
                    Rk = self._fv(self.Rw[k])
                ops.corr_fwd(lib, Lk, Rk, dsi, self.md, self.cstride, coff=c, u=(None if k == 6 else self.u[k]),
                             copy_left=True, zero_tail=True)
            x = ops.View(self.dsi[k], B, h, w, c + self.D + (0 if k == 6 else 1), ld)
            # 'mixed': the estimators of the three coarsest levels run plain bf16 in the forward pass too -- measured
            # contribution to the final disparity 8e-6 / 1e-5 / 1.3e-4 px (profiles/r02_precision_map.txt: rounding ONE group's
            # operands to bf16, everything else fp32), against 2.7e-3 px for level 3 and 7e-2 px for level 2, which keep
            # split-bf16 / exact fp32 like the pyramid and the context network
            fprec = 1 if (self.precision == "mixed" and k >= 4) else None
            for j, co in enumerate(EST):
                last = j == len(EST) - 1
                o = self._fv(self.V[k]) if last else self._fv(self.E[k][j])
                if last and k == 2 and self.sched.FUSE_HEAD and hasattr(lib, "conv2d_head"):
                    # the level-2 head also fills the disparity slot of the context network's input and seeds final = V2 + context7 (two copy
                    # launches on the critical chain before)
                    h4, w4, c4_ = self.fshape[4]
                    ops.conv2d_head(lib, x, self.W_(est_name(k, j + 1)), self.b_(est_name(k, j + 1)), o,
                                    copies=(ops.View(self.ctx_in, B, h4, w4, c4_ + 1, self.ctx_ld).slice(c4_, c4_ + 1), self._fv(self.final)))
                    head2_fused = True
                    x = o
                    continue
                if last and k != 2 and self._head_in_front(k, x):
                    # levels 6 .. 3: the head runs inside level k - 1's front-end launch, which needs its result first (Schedule.HEAD_IN_FRONT)
                    pending_head = (x, est_name(k, j + 1))
                    x = o
                    continue
                self._conv_fwd(lib, r, x, est_name(k, j + 1), o, alpha=(1.0 if last else ALPHA), precision=fprec,
                               shadow_consumer=(None if last else est_name(k, j + 2)))
                x = o
            if k != 2:
                sc = 2 ** (k - 1)
                if not self._front_fused():          # (fused: level k-1's front kernel computes u itself)
                    ops.resize_fwd(lib, self.V[k], self.u[k - 1], self.Hp // sc, self.Wp // sc, mul=20.0 / sc, mode=0)
                if k in make_disps and pending_head is None:
                    self._make_disp(lib, self.V[k], self.disp_k[k])
        assert pending_head is None
        # context network (MadNet._stereo_context_net, MadNet.py:122-171)
        h, w, c = self.fshape[4]
        cin = ops.View(self.ctx_in, B, h, w, c + 1, self.ctx_ld)
        concat_split = self.sched.FUSE_SPLITS and ctx_name(1) in self.banks32 and self.use_stream and self.partial_wgrad
        if concat_split:
            # the planes of tf.concat([left features, V2]) straight from the two sources: the first layer takes the planes, its streamed filter gradient
            # the hi plane, its input gradient has no mask -- nothing reads an fp32 copy of the concatenation
            key, pl = self._planes_of(cin)
            ops.plane_split(lib, [((self._half(self.F[4], False), self._fv(self.V[2])), pl)], self.dev, r.keep)
            self._fresh_planes.add(key); self._fresh.add(key)
        else:
            ops.copy_channels(lib, self._half(self.F[4], False), cin.slice(0, c))
            if not head2_fused:
                ops.copy_channels(lib, self._fv(self.V[2]), cin.slice(c, c + 1))
        x = cin
        for j, (co, rate) in enumerate(CTX[:-1]):
            o = self._fv(self.Cx[j])
            self._conv_fwd(lib, r, x, ctx_name(j + 1), o, dil=rate, shadow_consumer=ctx_name(j + 2))
            x = o
        # final_disp = V2_init + context7  (accumulating epilogue)
        if not head2_fused:
            ops.copy_channels(lib, self._fv(self.V[2]), self._fv(self.final))
        self._conv_acc(lib, x, ctx_name(7), self._fv(self.final), CTX[-1][1])
        if 2 in make_disps:
            self._make_disp(lib, self.final, self.disp_k[2])
        # rescaled_prediction: relu AFTER resize (MadNet.py:362-364)
        ops.resize_fwd(lib, self.final, self.pred, self.Hp, self.Wp, self.pt, self.pl, mul=-20.0, mode=2)
        self._stamp(lib, "forward_end")

    def _shadow(self, v, casts):
        """the bf16 shadow of View v (allocated on first use); queues its cast unless the producing kernel wrote it (self._fresh) or this
        batch already queued it"""
        key = (v.ptr, v.B, v.H, v.W, v.C)
        sh = self.shadows.get(key)
        if sh is None:
            sh = self.shadows[key] = ops.Shadow(v.B, v.H, v.W, v.C, self.dev)
        if key not in self._fresh and not any(c[1] is sh for c in casts):
            casts.append((v, sh))
        return sh

    def _out_shadow(self, v, consumer):
        """Shadow the PRODUCER of View v should write in its epilogue (mh_conv2d_sh), or None: only when the filter gradient of the layer
        `consumer` (a variable base name) is streamed in the plan being recorded."""
        if not (self.use_stream and self.fuse_shadows and self.partial_wgrad and ops._bwd_precision() == 1):
            return None
        if (consumer + "/weights") not in self._stream_train or v.npix < self.stream_min_pix:
            return None
        key = (v.ptr, v.B, v.H, v.W, v.C)
        sh = self.shadows.get(key)
        if sh is None:
            sh = self.shadows[key] = ops.Shadow(v.B, v.H, v.W, v.C, self.dev)
        self._fresh.add(key)
        return sh

    def _fresh_shadow(self, v):
        """the bf16 shadow of View v if a producer recorded earlier in this plan wrote it (the patch-staged input-gradient kernel then stages
        it instead of converting v), else None"""
        if not (self.sched.SHADOW_DGRAD and ops._bwd_precision() == 1):
            return None
        key = (v.ptr, v.B, v.H, v.W, v.C)
        return self.shadows.get(key) if key in self._fresh else None


    def _front_fused(self):
        return self.fuse_front and self.warping and self.cstride == 1 and self.D <= 9

    def _head_in_front(self, k, x):
        """the disparity head of level k inside level k - 1's front-end launch (Schedule.HEAD_IN_FRONT), when the library serves the shape"""
        if not (self.sched.HEAD_IN_FRONT and self._front_fused() and hasattr(self.lib, "level_front_head_ok")):
            return False
        h, w, c = self.fshape[FEAT[k - 1]]
        return self.lib.level_front_head_ok(x.H, x.W, h, w, c, x.C, self.md) == 1

    def _conv_acc(self, lib, x, base, out, rate):
        import ctypes as C
        w = self.W_(base)
        kh, kw, cin, cout = w.shape
        Ho, Wo, pt, pl = ops.conv_geometry(x.H, x.W, kh, kw, 1, rate)
        d = ops.conv_desc(x.B, x.H, x.W, Ho, Wo, cin, cout, kh, kw, 1, rate, pt, pl, 0, 0, x.ld, out.ld,
                          accumulate=1, alpha=1.0)
        lib.conv2d(C.byref(d), ops._p(x), ops._p(w), ops._p(self.b_(base)), ops._p(out), None, None)

    def _make_disp(self, lib, V, out):
        """MadNet._make_disp (MadNet.py:68-71): crop(resize(relu(-20 V)))."""
        ops.resize_fwd(lib, V, out, self.Hp, self.Wp, self.pt, self.pl, mul=-20.0, mode=1)

    def _flush_x0(self, r):
        """Schedule.IMAGE_CONV: the padding launch record_forward held back (it goes onto lane 1 with the loss reductions); called with the lane it should run on"""
        if getattr(self, "_x0_pending", False):
            ops.pad_reflect(r, self.lr, self.X0, self.pt, self.pl)
            self._x0_pending = False

    def record_loss_metrics(self, r, with_grad):
        """full-resolution reprojection loss (Stereo_Online_Adaptation.py:70) -- or, loss_kind 'proxy', the proxy-label
        mean_l1 of the continual variant (Stereo_Continual_Adaptation.py:75, weight 0.01) -- + EPE/bad3 (:74-82)."""
        side = self.sched.SIDE_LOSS and self.wgrad_lanes > 0 and hasattr(r, "lane")
        if not side:
            self._flush_x0(r)
        if self.loss_kind == "proxy":
            ops.proxy_loss(r, self.pred, self.proxy, self.proxy_ws, self.res_loss, self.dpred if with_grad else None, weight=0.01)
        elif side:
            # only the maps + the gradient are on the critical path; the reduction of the loss VALUE and the validation metrics
            # (read by the host after the step) run on a side lane next to the backward pass
            ops.reprojection_loss(r, self.left, self.right, self.pred, self.loss_ws, self.res_loss,
                                  self.dpred if with_grad else None, phase=1)
        else:
            ops.reprojection_loss(r, self.left, self.right, self.pred, self.loss_ws, self.res_loss,
                                  self.dpred if with_grad else None)
        if side:
            r.lane = 1
            try:
                self._stamp(r, "side_lane_first_op")
                self._flush_x0(r)               # (the padded frames: read by conv1's filter gradient only, on this lane, at the far end of the step)
                if self.loss_kind != "proxy":
                    ops.reprojection_loss(r, self.left, self.right, self.pred, self.loss_ws, self.res_loss, None, phase=2)
                ops.metrics(r, self.pred, self.gt, self.met_ws, self.res_met, 3.0)
            finally:
                r.lane = 0
        else:
            ops.metrics(r, self.pred, self.gt, self.met_ws, self.res_met, 3.0)

    # =========================================================================================
    # backward
    # =========================================================================================

    def _det_flush(self, lib, t, twin_all, base_all):
        """deterministic mode: t (a contiguous slice of base_all) += its fixed-point twin; recorded where the next reader of t follows"""
        if not self.deterministic:
            return
        import ctypes as _C
        off = (t.data_ptr() - base_all.data_ptr()) // 4
        assert t.is_contiguous() and 0 <= off and off + t.numel() <= base_all.numel()
        lib.det_flush(_C.c_void_p(t.data_ptr()), _C.c_void_p(twin_all.data_ptr() + 8 * off), t.numel(), None)

    def close(self):
        """deterministic mode: un-register this engine's ranges (the table holds 8 per process)"""
        import ctypes as _C
        for b in self._det_bases:
            self.lib.deterministic_remove(_C.c_void_p(b))
        self._det_bases = []

    def __del__(self):
        try:
            # (never from inside a stream capture: un-registering synchronises the device, which would invalidate the capture -- call close() explicitly)
            if self._det_bases and not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
                self.close()
        except Exception:
            pass

    def record_update(self, r, train_vars, lr, momentum=0.9, grad_scale=1.0, done=()):
        """MomentumOptimizer apply on the (coalesced) ranges of train_vars; done: sorted disjoint [first, end) ranges that
        record_backward(early_update=...) has updated already."""
        P = self.params

        def emit(a, b):
            ops.momentum(r, P.w[a:b], P.m[a:b], P.g[a:b], lr, momentum, grad_scale, n=b - a)
        for o, c in P.ranges(train_vars):
            a, end = o, o + c
            for d0, d1 in done:
                if d1 <= a:
                    continue
                if d0 >= end:
                    break
                if d0 > a:
                    emit(a, d0)
                a = max(a, d1)
            if a < end:
                emit(a, end)

    def record_update_adam(self, r, train_vars, lr, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
        """tf.train.AdamOptimizer(lr, 0.9).apply_gradients (Train.py:95,102) on the coalesced ranges of train_vars; the
        beta-power state advances once per step, after the last range."""
        P = self.params
        self._ensure_train_buffers()
        for o, c in P.ranges(train_vars):
            ops.adam(r, P.w[o:o + c], P.m[o:o + c], P.v[o:o + c], P.g[o:o + c], self.adam_state, lr, beta1, beta2, eps,
                     grad_scale, n=c)
        ops.adam_advance(r, self.adam_state, beta1, beta2)

    def _ensure_train_buffers(self):
        """Buffers only the offline-training plan needs: second Adam moment, beta powers, one gradient map and one result
        slot per predicted scale."""
        if getattr(self, "adam_state", None) is None:
            z = lambda *shape: torch.zeros(*shape, device=self.dev)
            self.params.v = z(self.params.total)
            self.adam_state = torch.tensor([0.9, 0.999], device=self.dev)
            self.ddisp_ms = {k: z(self.B, self.H0, self.W0) for k in LEVELS}
            self.res_loss_ms = z(6, 4)                 # rows: final, level 2 (context), 3, 4, 5, 6 = disparities[-1], [-2], ...
            self.sup_ws = z(self.lib.proxy_ws_floats(self.B, self.H0, self.W0))

    # =========================================================================================
    # compiled step plans
    # =========================================================================================
    def all_vars(self):
        return [n for n, _ in self.params.manifest]

    def pyramid_range(self):
        """(offset, count) of the pyramid's variables in the flat buffers: they lead the layout (madnet_manifest), the estimators and the
        context network follow -- the two pieces of the shared-model all-reduce."""
        rng = self.params.ranges([n for n, _ in self.params.manifest if "pyramid" in n])
        assert len(rng) == 1 and rng[0][0] == 0
        return rng[0]

    def build_plan(self, mode, lr=1e-4, block_vars=None, block_level=None, grad_scale=1.0, update=True,
                   blocks=None, part="all", loss_weights=None, max_disp=192.0, optimizer="momentum", momentum=0.9, collective=None, inputs=None):
        """mode: 'NONE' | 'FULL' | 'MAD' | 'TRAIN' (offline training step of Train.py: multi-scale supervised mean_l1 against
        self.gt with loss_weights from full to lowest resolution, every variable, Adam).
        For MAD: blocks = [(level, variable names), ...] (level in
        LEVELS, 2 = context output); block_level/block_vars is the single-block shorthand.
        optimizer: 'momentum' (Stereo_Online_Adaptation.py:122; `momentum` = its decay, 0.9 there) | 'adam' (the live demo, Demo/demo_model.py:164) for FULL / MAD.
        part: 'all' | 'grad' (everything up to the gradients) | 'update' (momentum apply only) -- the
        split lets a gradient all-reduce (shared-model multi-GPU mode) sit between two plans; 'grad_split' returns the 'grad' part as
        a LIST of two plans cut where the pyramid's backward pass starts (see madnet_manifest).
        collective: a madnet_hip.comm.Comm (shared-model mode, part='all', FULL / MAD): the gradient all-reduce is RECORDED between the backward pass and the
        optimizer (MH_OP_ALLREDUCE), so the step is one plan / one hipGraph.  FULL: [estimators + context + loss] leaves on a side lane where the pyramid's
        backward pass starts, [pyramid] follows behind it on lane 0; MAD: the block's ranges + the loss tail as one RCCL group.  Pass grad_scale = 1 / world.
        inputs: an ops.InputTable -- the plan's FIRST op fills left / right / gt / proxy from the device tensors the table names when the plan runs (mh_fetch_inputs:
        frames from a prefetcher's rotating slots without copy launches in front of a captured step); entries the host leaves empty keep the buffers as they are."""
        r = Recorder()
        if inputs is not None and part != "update":
            ops.fetch_inputs(r, inputs.ptr, [self.left, self.right, self.gt, self.proxy])
        self.wsa.reset()
        self._fresh = set()
        self.stamp_labels = []
        if mode in ("FULL", "TRAIN"):
            self._stream_train = set(self.all_vars())
        elif mode == "MAD":
            bl = blocks if blocks is not None else ([(block_level, block_vars)] if block_level is not None else [])
            self._stream_train = set(sum((list(bv) for _, bv in bl), [])) if len(bl) == 1 else set()      # several blocks re-run the backward: cast path
        else:
            self._stream_train = set()
        if part == "update":
            self._stream_train = set()
        # the filter-gradient split counts are resolved while the plan is recorded, from a process-wide hook another engine's recording may be scoping
        # (dispnet_engine.build_plan: 150 %): recordings are serialised
        lanes = self.wgrad_lanes
        if mode == "MAD" and self.mad_serial and lanes == 1:
            self.wgrad_lanes = 0
        try:
            with ops.TUNE_LOCK, ops.precision_scope(self.precision):
                if mode == "TRAIN":
                    return self._build_train_plan(r, lr, grad_scale, update, part, loss_weights, max_disp)
                return self._build_plan(r, mode, lr, block_vars, block_level, grad_scale, update, blocks, part, optimizer, momentum, collective)
        finally:
            self.wgrad_lanes = lanes

    def _build_train_plan(self, r, lr, grad_scale, update, part, loss_weights, max_disp):
        """Train.py:56-62,94-102: bulkhead off, loss = sum_i w_i * mean_l1(disparities[-(i+1)], gt, valid), Adam(lr, 0.9)."""
        self._ensure_train_buffers()
        lw = list(loss_weights) if loss_weights is not None else [1.0] * 10
        tv = self.all_vars()
        if part in ("all", "grad"):
            self.record_forward(r, make_disps=tuple(LEVELS))
            self._flush_x0(r)
            order = ["final"] + sorted(LEVELS)                      # disparities[-1], [-2] (context), [-3] (level 3) ... [-6] (level 6)
            heads = {}
            for i, hd in enumerate(order):
                pred = self.pred if hd == "final" else self.disp_k[hd]
                gbuf = self.dpred if hd == "final" else self.ddisp_ms[hd]
                ops.supervised_loss(r, pred, self.gt, self.sup_ws, self.res_loss_ms[i], gbuf, weight=lw[i], max_disp=max_disp)
                heads[hd] = gbuf
            ops.metrics(r, self.pred, self.gt, self.met_ws, self.res_met, 3.0)
            self.record_backward(r, None, tv, bulkhead=False, heads=heads)
        if update and part in ("all", "update"):
            self.record_update_adam(r, tv, lr, grad_scale=grad_scale)
        self._elide_fp32_gradient_maps(r)
        self._elide_fp32_activations(r)
        return r.compile()

    COLLECTIVE_LANE = 4            # the side lane of the shared-model step's first all-reduce (MH_MAX_LANES - 1: no filter-gradient batch uses it)

    def _build_plan(self, r, mode, lr, block_vars, block_level, grad_scale, update, blocks, part, optimizer="momentum", momentum=0.9, collective=None):
        if optimizer not in ("momentum", "adam"):
            raise ValueError("optimizer must be 'momentum' or 'adam'")
        if collective is not None and (part != "all" or mode not in ("FULL", "MAD") or not update):
            raise ValueError("collective= records the all-reduce inside a complete FULL / MAD step (part='all', update=True)")
        # one AdamOptimizer serves every train op of the demo graph (Demo/demo_model.py:164): per-variable slots, ONE pair of beta
        # powers that advances with every executed train op -- which is what record_update_adam does per call
        if optimizer == "momentum":
            record_update = lambda rr, tv_, lr_, grad_scale=1.0: self.record_update(rr, tv_, lr_, momentum=momentum, grad_scale=grad_scale)
        else:
            record_update = self.record_update_adam
        if blocks is None and block_level is not None:
            blocks = [(block_level, block_vars)]
        do_grad = part in ("all", "grad", "grad_split")
        do_upd = update and part in ("all", "update")
        if mode == "NONE":
            if do_grad:
                self.record_forward(r, need_x0=False)
                self.record_loss_metrics(r, with_grad=False)
        elif mode == "FULL":
            tv = self.all_vars()
            done = ()
            if do_grad:
                self.record_forward(r)
                self.record_loss_metrics(r, with_grad=True)
                eu = (lr, momentum, grad_scale) if (self.sched.EARLY_UPDATE and do_upd and part == "all" and optimizer == "momentum" and collective is None) else None
                at_cut = None
                if collective is not None:
                    P, lo = self.params, self.pyramid_range()[1]

                    def at_cut(rr):
                        # every gradient behind the pyramid's range + the loss result is final once the side lanes have joined: that range (73 % of the bytes)
                        # leaves on a lane of its own while the pyramid's backward pass runs on lane 0
                        lane0, nd0 = rr.lane, rr.nodefer
                        # (lane-to-lane edges: the all-reduce's lane waits for the filter-gradient lanes 1 .. 3 and is forked from lane 0 here; lane 0 itself goes on)
                        rr.join_lanes_next, rr.lane, rr.nodefer = 0b1110, self.COLLECTIVE_LANE, True
                        collective.allreduce(rr, [(P.g_loss, lo, P.total + 4 - lo)])
                        rr.lane, rr.nodefer = lane0, nd0
                cut_done = []
                if (collective is None and eu is None and self.sched.CUT_UPDATE and do_upd and part == "all" and optimizer == "momentum"
                        and self.wgrad_lanes > 0 and hasattr(r, "lane") and not self.deterministic):      # (the deterministic twin is flushed in front of the optimizer: one update)
                    P, lo = self.params, self.pyramid_range()[1]

                    def at_cut(rr):                              # noqa: F811
                        lane0, nd0 = rr.lane, rr.nodefer
                        variant = os.environ.get("MH_CUT_VARIANT", "lane4")
                        if variant == "lane0":                   # diagnostic: the update on lane 0 itself behind a join of the filter-gradient lanes
                            rr.join_lanes_next = 0b1110
                        elif variant == "lane4_defer":
                            rr.join_lanes_next, rr.lane, rr.nodefer = 0b1110, self.COLLECTIVE_LANE, False
                        else:
                            rr.join_lanes_next, rr.lane, rr.nodefer = 0b1110, self.COLLECTIVE_LANE, True
                        ops.momentum(rr, P.w[lo:P.total], P.m[lo:P.total], P.g[lo:P.total], lr, momentum, grad_scale, n=P.total - lo)
                        rr.lane, rr.nodefer = lane0, nd0
                        cut_done.append((lo, P.total))
                done = self.record_backward(r, "final", tv, bulkhead=False, early_update=eu, at_cut=at_cut)
                if cut_done:
                    done = tuple(done or ()) + tuple(cut_done)
                if collective is not None:
                    r.join_next = True                      # the pyramid's filter gradients (side lanes) and the first all-reduce (its lane) are behind us
                    collective.allreduce(r, [(self.params.g_loss, 0, self.pyramid_range()[1])])
                    r.join_next = True
            if do_upd:
                if done:
                    self.record_update(r, tv, lr, momentum=momentum, grad_scale=grad_scale, done=done)
                else:
                    record_update(r, tv, lr, grad_scale=grad_scale)
        elif mode == "MAD":
            if do_grad:
                self.record_forward(r, make_disps=tuple(lv for lv, _ in blocks))
                self.record_loss_metrics(r, with_grad=False)
            if do_grad and self.rscale != 1:
                if self.loss_kind == "proxy":
                    raise NotImplementedError("reprojectionScale != 1 is implemented for the reprojection loss (the online script)")
                ops.resize_image(r, self.left, self.left_s)           # inputs_modules (Stereo_Online_Adaptation.py:91-95)
                ops.resize_image(r, self.right, self.right_s)
            for lv, bv in blocks:
                if do_grad:
                    # loss of the block's prediction: reprojection (Stereo_Online_Adaptation.py:98-107) or, continual
                    # variant, proxy-label mean_l1 with weight 0.1 (Stereo_Continual_Adaptation.py:100-112)
                    if self.loss_kind == "proxy":
                        ops.proxy_loss(r, self.disp_k[lv], self.proxy, self.proxy_ws, self.res_loss_k, self.ddisp_k, weight=0.1)
                    elif self.rscale != 1:
                        Hs, Ws = self.H0 // self.rscale, self.W0 // self.rscale
                        ops.resize_fwd(r, self.disp_k[lv], self.p_s, Hs, Ws, mul=1.0, mode=0)
                        ops.reprojection_loss(r, self.left_s, self.right_s, self.p_s, self.loss_ws_s, self.res_loss_k, self.dp_s)
                        ops.resize_bwd(r, self.dp_s, self.disp_k[lv], self.ddisp_k, Hs, Ws, mul=1.0, mode=0)
                    else:
                        ops.reprojection_loss(r, self.left, self.right, self.disp_k[lv], self.loss_ws_k, self.res_loss_k,
                                              self.ddisp_k)
                    self.record_backward(r, lv, bv, bulkhead=True)
                if collective is not None:
                    # the block's gradient ranges + the loss tail (behind the gradient buffer) as ONE RCCL group between the block's backward pass and its update
                    P = self.params
                    rng = [(P.g_loss, o, c) for o, c in P.ranges(bv)]
                    if lv is blocks[0][0]:                          # (the loss result travels once per step: with the first block's gradients)
                        if rng and rng[-1][1] + rng[-1][2] == P.total:
                            rng[-1] = (P.g_loss, rng[-1][1], rng[-1][2] + 4)
                        else:
                            rng.append((P.g_loss, P.total, 4))
                    r.join_next = True
                    collective.allreduce(r, rng)
                    r.join_next = True
                if do_upd and part == "all":
                    record_update(r, bv, lr, grad_scale=grad_scale)
            if do_upd and part == "update":
                for lv, bv in blocks:
                    record_update(r, bv, lr, grad_scale=grad_scale)
        else:
            raise ValueError("unknown mode %r" % (mode,))
        self._stamp(r, "end")
        assert not getattr(self, "_x0_pending", False), "the padding launch record_forward held back was never recorded"
        self._elide_fp32_gradient_maps(r)
        self._elide_fp32_activations(r)
        return r.compile_parts() if part == "grad_split" else r.compile()

    # convenience: eager single forward -------------------------------------------------------
    def set_inputs(self, left, right, gt=None, proxy=None):
        self.left.copy_(torch.as_tensor(left, dtype=torch.float32).reshape(self.left.shape))
        self.right.copy_(torch.as_tensor(right, dtype=torch.float32).reshape(self.right.shape))
        if gt is not None:
            self.gt.copy_(torch.as_tensor(gt, dtype=torch.float32).reshape(self.gt.shape))
        if proxy is not None:
            self.proxy.copy_(torch.as_tensor(proxy, dtype=torch.float32).reshape(self.proxy.shape))
