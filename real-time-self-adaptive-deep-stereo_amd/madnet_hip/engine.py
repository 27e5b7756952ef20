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

PYR = [(3, 16, 2), (16, 16, 1), (16, 32, 2), (32, 32, 1), (32, 64, 2), (64, 64, 1),
       (64, 96, 2), (96, 96, 1), (96, 128, 2), (128, 128, 1), (128, 192, 2), (192, 192, 1)]
EST = [128, 128, 96, 64, 32, 1]
CTX = [(128, 1), (128, 2), (128, 4), (96, 8), (64, 16), (32, 1), (1, 1)]
LEVELS = (6, 5, 4, 3, 2)
FEAT = {6: 12, 5: 10, 4: 8, 3: 6, 2: 4}
ALPHA = 0.2   # MadNet._leaky_relu (Nets/MadNet.py:366-367)
# The reduction of the loss value + the validation metrics run on a side lane (SIDE_LOSS: 2.048 -> 2.030 ms since side launches are deferred,
# profiles/r02_experiments.txt #10, #20); the warp-gradient scatters on a lane of their own lost in every variant (2.056 / 2.27 ms) and
# stay in line.
ONE_FILL = True       # one zero fill for all level feature gradients + the g fill on the filter-gradient lane
FUSE_BACK = True     # one launch for a level's correlation gradient + warp gradient (mh_corr_warp_bwd)
PYR_BF16_FROM = int(os.environ.get("MH_PYR_BF16_FROM", "7"))     # 'mixed': pyramid layers from this one on run plain bf16 in the forward pass (13 = none)
# the first N filter-gradient batches of a backward pass are launched at once instead of after the next lane-0 op (MH_OP_NODEFER)
NODEFER_BATCHES = 0             # (module attribute: tests / experiments set it; early side launches measured slower, r03 #3)
SIDE_LOSS = True     # on since side launches are deferred: 2.048 -> 2.030 ms (r02z)


def _r4(c):
    return (c + 3) // 4 * 4


def pyr_name(i):
    return "model/gc-read-pyramid/conv%d" % i


def est_name(k, j):
    return "model/G%d/fgc-volume-filtering-%d/disp-%d" % (k, k, j)


def ctx_name(j):
    return "model/context-%d" % j


def _merge_ranges(ranges):
    out = []
    for a, b in sorted(ranges):
        if out and a <= out[-1][1]:
            out[-1] = (out[-1][0], max(out[-1][1], b))
        else:
            out.append((a, b))
    return out


def madnet_manifest(radius_d=2, stride=1):
    """Ordered [(variable name, shape)] -- flat-buffer order.  Names are the TF variable names of
    the reference graph."""
    D = 2 * radius_d // stride + 1
    out = []

    def conv(base, k, ci, co):
        out.append((base + "/weights", (k, k, ci, co)))
        out.append((base + "/biases", (co,)))

    # the twelve pyramid layers first, then estimator by estimator (the context network behind estimator 2): the backward pass
    # finishes the estimator / context gradients BEFORE it starts on the pyramid, so [estimators | loss] is one contiguous range whose
    # all-reduce overlaps the pyramid's backward pass in the shared-model mode, and the pyramid is the other (adapter.py).  A MAD
    # block = its pyramid layers (contiguous) + its estimator (contiguous): two ranges.
    for i in range(1, 13):
        conv(pyr_name(i), 3, PYR[i - 1][0], PYR[i - 1][1])
    for k in (2, 3, 4, 5, 6):
        cin = PYR[FEAT[k] - 1][1] + D + (0 if k == 6 else 1)
        for j, co in enumerate(EST):
            conv(est_name(k, j + 1), 3, cin, co)
            cin = co
        if k == 2:
            cin = PYR[3][1] + 1
            for j, (co, _) in enumerate(CTX):
                conv(ctx_name(j + 1), 3, cin, co)
                cin = co
    return out



# one launch for a head's output gradient + input gradient (mh_head_bwd) instead of resize gradient / copies + the K = 1 input-gradient kernel,
# and the level-2 head's forward pass storing its result in the context input and in `final` too (mh_conv2d_head)
# the first filter-gradient batches of a backward pass (context network, estimators 2 and 3: issued long before the step ends) run on 192
# workgroups instead of 256: a quarter less split workspace and a quarter of the CUs left to the main chain (profiles/r03_experiments.txt #20:
# 1.635 -> 1.629 ms; 128 / 96 workgroups: 1.649 / 1.652)
EARLY_WGS = 192
EARLY_BATCHES = 3
FUSE_HEAD = True
# FULL momentum steps: every filter-gradient batch is followed by the momentum update of its layers on its own lane, the launch behind the join covers
# only what is left.  OFF: prepared at the end of round 3 and never timed on the MI355X (profiles/r03_experiments.txt #26; bench.py --set engine.EARLY_UPDATE=True)
EARLY_UPDATE = False
# The step's tail (device time stamps, bench.py --stamps, round 4): the last filter-gradient batch (conv4 .. conv1) could only start behind the LAST input
# gradient -- conv1's filter gradient reads what conv2's input gradient writes -- so lane 0 sat idle for 91 us behind the chain (batch 76 us + join).
# TAIL_SPLIT: the filter gradients of conv4 .. conv2 (their operands are final one layer earlier) leave as a batch of their own BEFORE conv2's input
# gradient is launched and run beside it; only conv1's (the 3-channel image layer) is left for the tail.
PACK_SIDE = False     # mh_pack_weights on the side lane beside pad_reflect / conv1: FULL -3 us (noise), but NONE / MAD get a second stream: +55 / +35 us (r04_experiments.txt #19)
TAIL_SPLIT = False
# The flush behind the last input gradient: the streamed layers of the last batch (conv4 .. conv2) on lane 0 -- idle from there to the join -- while the
# side lane does the image layer (round 4: the side lane's second-to-last batch ends with the input-gradient chain, so the whole last batch, 62 us, was
# exposed: profiles/r04_experiments.txt #17)
TAIL_MAIN = True
# generalisation: the pyramid's filter gradients leave for the side lane in batches; a batch is flushed AFTER the input gradient of layer i for i in
# PYR_FLUSH_AFTER (the round-3 schedule: 9, 5, 1 = four layers per batch) and BEFORE the input gradient of layer i -- i.e. as soon as layer i's own
# filter gradient has its operands -- for i in PYR_FLUSH_BEFORE
PYR_FLUSH_AFTER = (9, 5, 1)
PYR_FLUSH_BEFORE = ()
# ... and that last batch (conv1's filter gradient + its split reduction) runs on a side lane of its own, so it starts the moment conv2's input gradient
# ends instead of queueing behind the conv4 .. conv2 batch on the filter-gradient lane (0 = same lane).  Its slice of the gradient buffer is zeroed on
# that lane too (first op of the backward pass): everything that touches those floats stays in ONE lane's order.
TAIL_LANE = 2
# 'mixed': the split-bf16 forward layers (stride-1 3x3, > bank_small_maxpix pixels) run from PRE-SPLIT operands (mh_conv2d_planes, csrc/conv_planes.hip):
# activations as hi / lo bf16 planes -- hi is the shadow the backward pass reads anyway -- written by the producer's epilogue, staged by LDS DMA
# Deterministic test mode: the float atomics of a step (bias gradients, warp-gradient scatter) accumulate into
# 64-bit fixed-point twins (mh_deterministic_add) that the plan flushes behind every level's scatter and in front of the optimizer -- two replays
# of the same step then give bit-identical weights.  At most four deterministic engines per process (two ranges each).
DETERMINISTIC = os.environ.get("MH_DETERMINISTIC", "0") == "1"
USE_PLANES = os.environ.get("MH_CONV_PLANES", "1") != "0"
# ... and the fp32 copy of such an activation is not stored when no op of the plan reads it (engine._elide_fp32_activations)
PLANES_ONLY = True
# tests: fill every fp32 buffer whose store a plan elides with NaN when the plan is built (engine._note_elided)
POISON_ELIDED = os.environ.get("MH_POISON_ELIDED", "0") == "1"
# ... and the planes of tensors no plane kernel produces are written by THEIR producers (the level front end, the exact-fp32 layers in front of conv4 /
# conv6, one concat-split for the context network's input) instead of by a split launch in front of every consumer
FUSE_SPLITS = True
# ... and the INPUT GRADIENTS of those layers (and of the 1/8-resolution estimator's) run the same kernel with one plane (mh_conv2d_planes_bwd): dz from the
# bf16 shadow its producer wrote, the leaky mask from the activation's hi plane, the result as a shadow (+ fp32 only where something reads it)
PLANES_DGRAD = True
# diagnostics (bench.py --stamps): device time stamps (mh_stamp) recorded as plan ops at the start of the step, the end of the forward pass, the first
# op of the side lane, the start / end of every filter-gradient batch, the end of the input-gradient chain, the join and the end of the step --
# the REPLAYED graph timed from the inside, without a tracer.  Each stamp is a one-lane kernel: the stamped plan is a few us slower than the plain one.
STAMPS = False
# input gradients stage the bf16 shadow of dz when the previous input gradient's epilogue wrote one (mh_conv2d_sh2)
SHADOW_DGRAD = True
# ... and then do not store the fp32 gradient map at all when its only reader is such an input gradient (engine._elide_fp32_gradient_maps)
SHADOW_ONLY = True


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


class MadNetEngine(object):
    def __init__(self, lib, H, W, B=1, device="cuda", radius_d=2, stride=1, warping=True, weights=None, precision="fp32"):
        """precision: 'fp32' = exact fp32 MFMA (parity path, default) | 'bf16' = bf16 MFMA inputs with fp32
        accumulation in the conv forward / input-gradient kernels (throughput mode; tensors stay fp32)."""
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
        self.use_planes = self.use_bank and precision == "mixed" and USE_PLANES
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
        self.deterministic = DETERMINISTIC
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
        if self.precision == "mixed" and i >= PYR_BF16_FROM:
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
        for i in range(2, 13):                                      # the stride-1 pyramid layers and the small stride-2 ones (conv7 / 9 / 11)
            h, w = (self.fshape[i][0], self.fshape[i][1])
            if PYR[i - 1][2] == 2:
                if 2 * B * h * w > self.bank_small_maxpix:
                    continue
                stride2.add(pyr_name(i))
            layers.append((pyr_name(i), self._pyr_code(i, fcode), 2 * B * h * w))
        for k in LEVELS:
            h, w, _ = self.fshape[FEAT[k]]
            code = 1 if (self.precision == "mixed" and k >= 4) else fcode
            layers += [(est_name(k, j), code, B * h * w) for j in range(1, 7)]
        h, w, _ = self.fshape[4]
        layers += [(ctx_name(j), fcode, B * h * w) for j in range(1, 8)]
        plan = []
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
        if bcode == 1 and self.use_planes and PLANES_DGRAD:
            for i in range(3, 13):
                h, w = self.fshape[i][0], self.fshape[i][1]
                if PYR[i - 1][2] == 2 and 2 * B * h * w > self.bank_small_maxpix and (i - 1) not in FEAT.values():
                    _, _, K, N = shapes[pyr_name(i) + "/weights"]
                    dxv = self._fv(self.dF[i - 1])
                    if ops.conv2d_planes_bwd_ok(self.lib, dxv, self.W_(pyr_name(i)), 1, stride=2):
                        plan.append((pyr_name(i), 1, 3))
        return plan

    def _stamp(self, lib, label):
        if not STAMPS or not hasattr(lib, "stamp"):
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
        if not (self.use_planes and PLANES_DGRAD):
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
        wb32 = self.banks32.get(base) if stride == 1 else None
        if wb32 is not None:
            xp = self._in_planes(lib, x, r)
            key, op_ = self._planes_of(o)
            ops.conv2d_planes(lib, xp, self.W_(base), wb32, self.b_(base), out=o, out_planes=op_, dil=dil, alpha=alpha)
            self._fresh_planes.add(key)
            self._fresh.add(key)
            return
        if shadow_consumer and shadow_consumer in self.banks32 and FUSE_SPLITS:
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

    def record_forward(self, r, make_disps=()):
        B, lib = self.B, r
        head2_fused = False
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
            side_pack = PACK_SIDE and hasattr(lib, "lane") and self.wgrad_lanes > 0
            if side_pack:
                lib.lane = 1
            try:
                ops.pack_weights(lib, [(self.W_(n), self._bank_of(trans)[n], planes, trans) for n, planes, trans in plan],
                                 self.dev, r.keep)
            finally:
                if side_pack:
                    lib.lane = 0
        ops.pad_reflect(lib, self.lr, self.X0, self.pt, self.pl)
        x = ops.View(self.X0, 2 * B, self.Hp, self.Wp, 3, 4)
        for i, (ci, co, s) in enumerate(PYR, 1):
            o = self._fv(self.F[i])
            if i == 2 and self.use_bank and PACK_SIDE and hasattr(lib, "lane") and self.wgrad_lanes > 0:
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
            if fused:
                # u_k = resize(V_{k+1}) * 20 / 2^k (MadNet.py:274), warp, cost volume + concat: one launch
                xin = ops.View(self.dsi[k], B, h, w, c + self.D + 1, ld)
                pl = None
                if FUSE_SPLITS and self.use_planes and self.cstride == 1:
                    key, pl_ = self._planes_of(xin)
                    if est_name(k, 1) in self.banks32:
                        pl = pl_                                  # hi + lo: the estimator's first layer runs from planes
                        self._fresh_planes.add(key); self._fresh.add(key)
                    elif (est_name(k, 1) + "/weights") in self._stream_train and self.use_stream and self.partial_wgrad and ops._bwd_precision() == 1:
                        pl = pl_.hi                               # hi only: the shadow its streamed filter gradient reads (no cast in the backward pass)
                        self._fresh.add(key)
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
                if last and k == 2 and FUSE_HEAD and hasattr(lib, "conv2d_head"):
                    # the level-2 head also fills the disparity slot of the context network's input and seeds final = V2 + context7 (two copy
                    # launches on the critical chain before)
                    h4, w4, c4_ = self.fshape[4]
                    ops.conv2d_head(lib, x, self.W_(est_name(k, j + 1)), self.b_(est_name(k, j + 1)), o,
                                    copies=(ops.View(self.ctx_in, B, h4, w4, c4_ + 1, self.ctx_ld).slice(c4_, c4_ + 1), self._fv(self.final)))
                    head2_fused = True
                    x = o
                    continue
                self._conv_fwd(lib, r, x, est_name(k, j + 1), o, alpha=(1.0 if last else ALPHA), precision=fprec,
                               shadow_consumer=(None if last else est_name(k, j + 2)))
                x = o
            if k != 2:
                sc = 2 ** (k - 1)
                if not self._front_fused():          # (fused: level k-1's front kernel computes u itself)
                    ops.resize_fwd(lib, self.V[k], self.u[k - 1], self.Hp // sc, self.Wp // sc, mul=20.0 / sc, mode=0)
                if k in make_disps:
                    self._make_disp(lib, self.V[k], self.disp_k[k])
        # context network (MadNet._stereo_context_net, MadNet.py:122-171)
        h, w, c = self.fshape[4]
        cin = ops.View(self.ctx_in, B, h, w, c + 1, self.ctx_ld)
        concat_split = FUSE_SPLITS and ctx_name(1) in self.banks32 and self.use_stream and self.partial_wgrad
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
        if not (SHADOW_DGRAD and ops._bwd_precision() == 1):
            return None
        key = (v.ptr, v.B, v.H, v.W, v.C)
        return self.shadows.get(key) if key in self._fresh else None

    def _elide_fp32_gradient_maps(self, r):
        """Post-pass over the recorded plan (dead-store elimination): an input-gradient launch that writes the bf16 shadow of its result does not
        store the fp32 map when the ONLY op that touches that buffer afterwards is the next input gradient and that launch stages the shadow
        (mh_conv2d_takes_shadows answers for the recorded descriptor): inside the 1/4-resolution estimator and the context network the gradient
        maps then exist in bf16 only (15.7 MB less written per 128-channel layer)."""
        if not (SHADOW_DGRAD and SHADOW_ONLY):
            return 0
        import ctypes as C
        from . import _ffi
        ops_ = r.ops
        n = 0
        spans = []
        for idx, o in enumerate(ops_):
            if o.kind == _ffi.OP_CONV and o.i[13] == 1 and o.p[7] and not o.i[18] and o.i[22] == 1 and o.p[3]:
                spans.append((idx, int(o.p[3]), int(o.p[3]) + 4 * o.i[0] * o.i[3] * o.i[4] * o.i[16]))
        for idx, lo, hi in spans:
            if any(a < hi and lo < a + nb for _, a, nb in getattr(r, "refs", ())):
                continue                # a device table (cast / split segment) reads the map
            users = []
            for j, q in enumerate(ops_):
                if j == idx:
                    continue
                if any(q.p[k] and lo <= int(q.p[k]) < hi for k in range(8)):
                    users.append(j)
            if len(users) != 1 or users[0] < idx:
                continue
            c = ops_[users[0]]
            if not (c.kind == _ffi.OP_CONV and c.i[13] == 1 and int(c.p[0]) == lo and (c.i[23] & 1) and c.i[22] == 1):
                continue
            if sum(1 for k in range(8) if c.p[k] and lo <= int(c.p[k]) < hi) != 1:
                continue
            if not (self._takes_shadows(c) & 1):
                continue
            ops_[idx].i[23] |= 4
            c.i[23] |= 8            # MH_CONV_IN_F32_STALE: a replay whose dispatch no longer stages the shadow is refused, not wrong
            self._note_elided(r, lo, hi - lo)
            n += 1
        return n

    def _note_elided(self, r, ptr, nbytes):
        """An fp32 buffer no op of this plan writes any more.  Kept on the recorder / plan (plan.elided) so that a reader OUTSIDE the plan can ask; with
        MH_POISON_ELIDED=1 (tests) the buffer is filled with NaN at once: an op that still reads it -- a device-table op whose table building forgot
        Recorder.note_refs, a debug read of engine.E / Cx -- then fails loudly instead of consuming a stale map."""
        if not hasattr(r, "elided"):
            r.elided = []
        r.elided.append((int(ptr), int(nbytes)))
        if POISON_ELIDED:
            t = self._tensor_by_ptr().get(int(ptr))
            if t is not None:
                t.fill_(float("nan"))

    def _tensor_by_ptr(self):
        out = {}
        for k in LEVELS:
            for t in list(self.E[k]) + list(self.dE[k]):
                out[t.data_ptr()] = t
        for t in list(self.Cx) + list(self.dCx):
            out[t.data_ptr()] = t
        return out

    def _takes_shadows(self, c):
        """mh_conv2d_takes_shadows for a recorded OP_CONV: bit 1 = the launch stages in_shadow, bit 2 = it reads the mask from mask_shadow"""
        import ctypes as C
        from . import _ffi
        d = _ffi.ConvDesc(*([c.i[k] for k in range(18)] + [c.i[18], c.f[0], c.f[1], c.i[19], c.i[20], c.i[22]]))
        return self.lib.conv2d_takes_shadows(C.byref(d), C.c_void_p(c.p[0]), C.c_void_p(c.p[1]), C.c_void_p(c.p[6]), C.c_void_p(c.p[3]), C.c_void_p(c.p[4]))

    def _standalone_activations(self):
        """{data pointer: bytes} of the activation tensors that are allocations of their own (no view of them can start in front of them): the only
        candidates for an elided fp32 store"""
        out = {}
        for k in LEVELS:
            for t in self.E[k]:
                out[t.data_ptr()] = t.numel() * 4
        for t in self.Cx:
            out[t.data_ptr()] = t.numel() * 4
        for k in LEVELS:                    # ... and the gradient maps between the input gradients of an estimator / the context network
            for t in self.dE[k]:
                out[t.data_ptr()] = t.numel() * 4
        for t in self.dCx:
            out[t.data_ptr()] = t.numel() * 4
        return out

    def _elide_fp32_activations(self, r):
        """Post-pass (dead-store elimination, forward side): a plane-writing forward layer (OP_CONV_PLANES) does not store its fp32 result when no op
        of the recorded plan reads that tensor -- the next forward layer takes the planes, the filter gradient the hi plane, the input gradient of the
        next layer the sign of the hi plane for its leaky mask (it gets MH_CONV_MASK_F32_STALE, so a replay under another dispatch fails loudly).
        Readers are found conservatively: any pointer field of any op, and any tensor a device table of an op references (Recorder.refs), that
        OVERLAPS the buffer."""
        if not (self.use_planes and PLANES_ONLY):
            return 0
        from . import _ffi
        ops_ = r.ops
        cand = self._standalone_activations()
        n = 0
        for idx, o in enumerate(ops_):
            if o.kind == _ffi.OP_CONV_PLANES and o.p[4] and o.p[5] and o.p[6]:
                slot = 4                    # forward: fp32 result beside both planes
            elif o.kind == _ffi.OP_CONV_PLANES_BWD and o.p[3] and o.p[4]:
                slot = 3                    # input gradient: fp32 map beside its shadow
            else:
                continue
            lo = int(o.p[slot])
            if lo not in cand:
                continue
            hi = lo + cand[lo]
            if any(a < hi and lo < a + nb for _, a, nb in getattr(r, "refs", ())):
                continue
            ok, mask_users = True, []
            for j, q in enumerate(ops_):
                if j == idx:
                    continue
                hits = [k for k in range(8) if q.p[k] and lo <= int(q.p[k]) < hi]
                if not hits:
                    continue
                # the only tolerated readers: an input gradient (tiled families) that was given this tensor as its leaky mask TOGETHER with the mask's
                # shadow and whose kernel tests the shadow -- or as its dz together with dz's shadow and whose kernel stages the shadow
                if (q.kind == _ffi.OP_CONV and q.i[13] == 1 and hits == [4] and int(q.p[4]) == lo and (q.i[23] & 2) and q.i[22] == 1
                        and (self._takes_shadows(q) & 2)):
                    mask_users.append((q, 16))          # MH_CONV_MASK_F32_STALE
                    continue
                if (q.kind == _ffi.OP_CONV and q.i[13] == 1 and hits == [0] and int(q.p[0]) == lo and (q.i[23] & 1) and q.i[22] == 1
                        and (self._takes_shadows(q) & 1)):
                    mask_users.append((q, 8))           # MH_CONV_IN_F32_STALE
                    continue
                ok = False
                break
            if not ok:
                continue
            o.p[slot] = None
            for q, bit in mask_users:
                q.i[23] |= bit
            self._note_elided(r, lo, hi - lo)
            n += 1
        return n

    def _front_fused(self):
        return self.fuse_front and self.warping and self.cstride == 1 and self.D <= 9

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

    def record_loss_metrics(self, r, with_grad):
        """full-resolution reprojection loss (Stereo_Online_Adaptation.py:70) -- or, loss_kind 'proxy', the proxy-label
        mean_l1 of the continual variant (Stereo_Continual_Adaptation.py:75, weight 0.01) -- + EPE/bad3 (:74-82)."""
        side = SIDE_LOSS and self.wgrad_lanes > 0 and hasattr(r, "lane")
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

    def record_backward(self, r, head, train_vars, bulkhead, heads=None, early_update=None):
        """early_update = (lr, momentum, grad_scale) (EARLY_UPDATE, FULL momentum steps): the update of a batch's layers follows the batch's reduction on its
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
        g_side = ONE_FILL and self.wgrad_lanes == 1 and hasattr(lib, "lane")
        tail_vars = [pyr_name(1) + "/weights", pyr_name(1) + "/biases"]
        tail_lane = TAIL_LANE if (TAIL_SPLIT and TAIL_LANE and g_side and pyr_tr[1] and pyr_tr[2] and all(v in train_vars for v in tail_vars)) else 0
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
        prezero = ONE_FILL and self.warping and not bulkhead
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
            tail (the flush behind the LAST input gradient, TAIL_MAIN): nothing follows on lane 0 any more, so the batch is split -- the layers of the
            streamed kernel (conv4 .. conv2) run on lane 0 itself while the side lane does the image layer's gradient and its reduction."""
            if not pending:
                return
            # (not with early_update: the ranges a batch completes are collected per flush, not per half)
            if (tail and TAIL_MAIN and early_update is None and self.wgrad_lanes > 0 and hasattr(lib, "lane") and self.use_stream and self.partial_wgrad
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
                lib.nodefer = nflush[0] < NODEFER_BATCHES        # the first batches (context network, 1/4-resolution estimator) carry most of the work
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
                                     target_wgs=(EARLY_WGS if (EARLY_WGS and self.B == 1 and nflush[0] <= EARLY_BATCHES) else None))
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
            if not (FUSE_HEAD and hasattr(lib, "head_bwd") and up_V[k] and ("V", k) not in written):
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
            elif FUSE_BACK and ("F", f, 1) in written:
                # the level's correlation + concat gradient and the warp gradient in ONE launch (mh_corr_warp_bwd): the gradient w.r.t. the warped
                # features never goes to memory; the scatter target was zeroed by the pass's single fill (or holds earlier contributions)
                du = self.du[k] if need_u else None
                ops.corr_warp_bwd(lib, g, Lk, self._fv(self.Rw[k]), self._half(self.F[f], True), self.u[k], dL, self._half(self.dF[f], True), du,
                                  self.md, self.cstride, coff=c, acc_l=acc_flag(("F", f, 0)), copy_left=True)
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
                if (TAIL_SPLIT and i == 2) or i in PYR_FLUSH_BEFORE:
                    flush()                 # (TAIL_SPLIT: conv4 .. conv2 beside conv2's input gradient, not behind it)
                if need_dx:
                    # dF[i-1] is complete after this launch (the cost-volume contributions were written earlier): it is the dz of layer i - 1
                    sh = self._out_shadow(self._fv(self.dF[i - 1]), pyr_name(i - 1)) if (i - 1 > 1) else None       # (conv1's 3-channel input keeps the tiled kernel)
                    wbt = self.banks32t.get(pyr_name(i)) if not accumulate else None       # (stride-2 layers: only those _bank_plan gave a bank)
                    dzs = self._fresh_shadow(self._fv(self.dF[i])) if wbt is not None else None
                    mks = self._fresh_shadow(self._fv(self.F[i - 1])) if wbt is not None else None
                    if wbt is not None and dzs is not None and mks is not None and sh is not None:
                        ops.conv2d_planes_bwd(lib, dzs, self.W_(pyr_name(i)), wbt, dx=self._fv(self.dF[i - 1]), dx_shadow=sh, mask_shadow=mks, mask_alpha=ALPHA,
                                              stride=PYR[i - 1][2])
                        if i in PYR_FLUSH_AFTER:
                            flush(lane=(tail_lane if i == 1 else None))
                        continue
                    ops.conv2d_dgrad(lib, self._fv(self.dF[i]), self.W_(pyr_name(i)), self._fv(self.dF[i - 1]),
                                     stride=PYR[i - 1][2], accumulate=accumulate, mask_ref=self._fv(self.F[i - 1]),
                                     mask_alpha=ALPHA, wb=self.Wd_(pyr_name(i)), shadow=sh,
                                     dz_shadow=self._fresh_shadow(self._fv(self.dF[i])), mask_shadow=self._fresh_shadow(self._fv(self.F[i - 1])))
                if i in PYR_FLUSH_AFTER:
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
                   blocks=None, part="all", loss_weights=None, max_disp=192.0, optimizer="momentum", momentum=0.9):
        """mode: 'NONE' | 'FULL' | 'MAD' | 'TRAIN' (offline training step of Train.py: multi-scale supervised mean_l1 against
        self.gt with loss_weights from full to lowest resolution, every variable, Adam).
        For MAD: blocks = [(level, variable names), ...] (level in
        LEVELS, 2 = context output); block_level/block_vars is the single-block shorthand.
        optimizer: 'momentum' (Stereo_Online_Adaptation.py:122; `momentum` = its decay, 0.9 there) | 'adam' (the live demo, Demo/demo_model.py:164) for FULL / MAD.
        part: 'all' | 'grad' (everything up to the gradients) | 'update' (momentum apply only) -- the
        split lets a gradient all-reduce (shared-model multi-GPU mode) sit between two plans; 'grad_split' returns the 'grad' part as
        a LIST of two plans cut where the pyramid's backward pass starts (see madnet_manifest)."""
        r = Recorder()
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
                return self._build_plan(r, mode, lr, block_vars, block_level, grad_scale, update, blocks, part, optimizer, momentum)
        finally:
            self.wgrad_lanes = lanes

    def _build_train_plan(self, r, lr, grad_scale, update, part, loss_weights, max_disp):
        """Train.py:56-62,94-102: bulkhead off, loss = sum_i w_i * mean_l1(disparities[-(i+1)], gt, valid), Adam(lr, 0.9)."""
        self._ensure_train_buffers()
        lw = list(loss_weights) if loss_weights is not None else [1.0] * 10
        tv = self.all_vars()
        if part in ("all", "grad"):
            self.record_forward(r, make_disps=tuple(LEVELS))
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

    def _build_plan(self, r, mode, lr, block_vars, block_level, grad_scale, update, blocks, part, optimizer="momentum", momentum=0.9):
        if optimizer not in ("momentum", "adam"):
            raise ValueError("optimizer must be 'momentum' or 'adam'")
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
                self.record_forward(r)
                self.record_loss_metrics(r, with_grad=False)
        elif mode == "FULL":
            tv = self.all_vars()
            done = ()
            if do_grad:
                self.record_forward(r)
                self.record_loss_metrics(r, with_grad=True)
                eu = (lr, momentum, grad_scale) if (EARLY_UPDATE and do_upd and part == "all" and optimizer == "momentum") else None
                done = self.record_backward(r, "final", tv, bulkhead=False, early_update=eu)
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
                if do_upd and part == "all":
                    record_update(r, bv, lr, grad_scale=grad_scale)
            if do_upd and part == "update":
                for lv, bv in blocks:
                    record_update(r, bv, lr, grad_scale=grad_scale)
        else:
            raise ValueError("unknown mode %r" % (mode,))
        self._stamp(r, "end")
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


def ops_fill(lib, t, off, count):
    """record/launch a zero fill of t.flatten()[off:off+count]."""
    import ctypes as C
    flat = t.reshape(-1)
    lib.fill(C.c_void_p(flat.data_ptr() + 4 * off), count, 0.0, None)
