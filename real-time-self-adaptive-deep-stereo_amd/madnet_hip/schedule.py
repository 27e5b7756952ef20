"""How a step is scheduled -- the A/B switches of the plan recorders, as IMMUTABLE values carried by the engine that records with them.

Rounds 1-4 kept these as module attributes of madnet_hip/engine.py (tests and bench.py --set assigned to them): two engines of one process saw each
other's experiment state, and a plan did not know what it had been recorded with.  Now:

    eng = MadNetEngine(lib, H, W, ..., schedule=Schedule(FUSE_HEAD=False))        # this engine only
    eng.sched.FUSE_HEAD                                                            # frozen: eng.sched.FUSE_HEAD = True raises
    dataclasses.replace(eng.sched, TAIL_MAIN=False)                                # a variant for another engine

Defaults = the schedule of the committed bench line; a field whose default comes from the environment reads it when the Schedule is CREATED.
Every measured claim cites profiles/rNN_experiments.txt."""
import os
from dataclasses import dataclass, field


def _env_flag(name, default):
    return lambda: os.environ.get(name, default) != "0"


@dataclass(frozen=True)
class Schedule(object):
    """MadNetEngine.record_forward / record_backward / build_plan"""
    # The reduction of the loss value + the validation metrics run on a side lane (2.048 -> 2.030 ms since side launches are deferred,
    # r02_experiments.txt #10, #20); the warp-gradient scatters on a lane of their own lost in every variant (2.056 / 2.27 ms) and stay in line.
    SIDE_LOSS: bool = True
    # (round 6) pyramid layers (1-based) whose STRIDE-2 forward pass runs the stride-2 plane kernel (conv_planes_s2fwd_kernel: split-bf16 from the producer's
    # planes) instead of the exact-fp32 tiled kernel.  conv5 reads conv4's planes (a plane kernel wrote them); conv3 would need conv2's lo plane (one more launch)
    PLANES_S2_FWD: tuple = (5,)
    # stride-2 input gradients whose target already holds a contribution (conv5 -> the level-2 features) on the accumulating parity-class kernel: measured +6 us per step
    # (12.4 us against 12.9 us for the launch, 1.3296 / 1.3283 vs 1.3239 / 1.3225 ms for the step: r06_experiments.txt #9) -- off
    PLANES_S2_ACC: bool = False
    # (round 6) the fused back end of a level (mh_corr_warp_bwd) is the FIRST writer of its level's feature gradient: no zero fill of the 13.9 MB of level
    # feature gradients at the head of the backward pass, no read of the halves it writes (the row-owned kernel gathers -- it never needed zeros to add to)
    FIRST_WRITER: bool = True
    ONE_FILL: bool = True            # one zero fill for all level feature gradients + the g fill on the filter-gradient lane
    FUSE_BACK: bool = True           # one launch for a level's correlation gradient + warp gradient (mh_corr_warp_bwd)
    # 'mixed': pyramid layers from this one on run plain bf16 in the forward pass (13 = none)
    PYR_BF16_FROM: int = field(default_factory=lambda: int(os.environ.get("MH_PYR_BF16_FROM", "7")))
    # the first N filter-gradient batches of a backward pass are launched at once instead of after the next lane-0 op (MH_OP_NODEFER); early side
    # launches measured slower (r03 #3)
    NODEFER_BATCHES: int = 0
    # the first filter-gradient batches of a backward pass (context network, estimators 2 and 3: issued long before the step ends) run on 192
    # workgroups instead of 256: a quarter less split workspace and a quarter of the CUs left to the main chain (r03 #20: 1.635 -> 1.629 ms;
    # 128 / 96 workgroups: 1.649 / 1.652)
    EARLY_WGS: int = 192
    EARLY_BATCHES: int = 3
    # one launch for a head's output gradient + input gradient (mh_head_bwd) instead of resize gradient / copies + the K = 1 input-gradient kernel,
    # and the level-2 head's forward pass storing its result in the context input and in `final` too (mh_conv2d_head)
    FUSE_HEAD: bool = True
    # the disparity heads of levels 6 .. 3 run INSIDE the next level's front-end launch (mh_level_front_head_fwd) instead of as launches of their own in front
    # of it: four 4.5 - 5 us nodes off the forward chain (r05_experiments.txt #11)
    HEAD_IN_FRONT: bool = field(default_factory=_env_flag("MH_HEAD_IN_FRONT", "1"))
    # conv1 (3 -> 16, stride 2) runs straight from the frames through the reflection (mh_conv_image_fwd: exact fp32, one thread per output pixel) instead of
    # mh_pad_reflect + the row-streaming kernel: 29 us -> ~10 us at the head of the forward chain.  The padded copy X0 is then only read by conv1's filter
    # gradient at the far end of the step: its padding launch leaves on lane 1 (or not at all when the plan has no backward pass)
    IMAGE_CONV: bool = field(default_factory=_env_flag("MH_IMAGE_CONV", "1"))
    # FULL momentum steps: every filter-gradient batch is followed by the momentum update of its layers on its own lane, the launch behind the join
    # covers only what is left.  Measured worse together with the tail split (r04 #17)
    EARLY_UPDATE: bool = False
    # (round 6) FULL momentum steps of a private model: where the pyramid's backward pass starts, the estimators' and the context network's gradients are final -- 73 % of the
    # parameters, one contiguous range behind the pyramid's -- and their update leaves on a lane of its own (it waits for the filter-gradient lanes, lane 0 walks on);
    # the launch behind the final join covers the pyramid's range only.  ONE more launch (EARLY_UPDATE: seven), off the step's tail.  Measured +8 us (r6ag): off
    CUT_UPDATE: bool = False
    # mh_pack_weights on the side lane beside pad_reflect / conv1: FULL -3 us (noise), but NONE / MAD get a second stream: +55 / +35 us (r04 #19)
    PACK_SIDE: bool = False
    # The step's tail (device time stamps, bench.py --stamps, round 4): the last filter-gradient batch (conv4 .. conv1) could only start behind the
    # LAST input gradient -- conv1's filter gradient reads what conv2's input gradient writes -- so lane 0 sat idle for 91 us behind the chain.
    # TAIL_SPLIT: the filter gradients of conv4 .. conv2 leave as a batch of their own BEFORE conv2's input gradient is launched (measured worse).
    TAIL_SPLIT: bool = False
    # The flush behind the last input gradient: the streamed layers of the last batch (conv4 .. conv2) on lane 0 -- idle from there to the join --
    # while the side lane does the image layer (r04 #17)
    TAIL_MAIN: bool = True
    # the pyramid's filter gradients leave for the side lane in batches; a batch is flushed AFTER the input gradient of layer i for i in
    # PYR_FLUSH_AFTER (9, 5, 1 = four layers per batch) and BEFORE the input gradient of layer i for i in PYR_FLUSH_BEFORE
    PYR_FLUSH_AFTER: tuple = (9, 5, 1)
    # the estimators' filter gradients leave as one batch per level in this tuple (level 6 always flushes); a level not named rides with the next batch.
    # Every flush is a fork edge on lane 0 (a 4.5 us gap in front of the level's correlation gradient in the traced graph), yet fewer batches measured
    # SLOWER: (2,3,6) +5 us, (2,6) +17 us, (6,) +190 us per step (r05_experiments.txt #12)
    EST_FLUSH_AFTER: tuple = (2, 3, 4, 5, 6)
    PYR_FLUSH_BEFORE: tuple = ()
    # (TAIL_SPLIT) the last batch on a side lane of its own (0 = same lane); its slice of the gradient buffer is zeroed on that lane too
    TAIL_LANE: int = 2
    # Deterministic test mode: the float atomics of a step (bias gradients; the warp-gradient scatter when the atomic form of
    # mh_corr_warp_bwd runs) accumulate into 64-bit fixed-point twins (mh_deterministic_add) that the plan flushes in front of their readers --
    # two replays of the same step give bit-identical weights.  At most four deterministic engines per process (two ranges each).
    DETERMINISTIC: bool = field(default_factory=lambda: os.environ.get("MH_DETERMINISTIC", "0") == "1")
    # 'mixed': the split-bf16 forward layers (stride-1 3x3, > bank_small_maxpix pixels) run from PRE-SPLIT operands (mh_conv2d_planes):
    # activations as hi / lo bf16 planes -- hi is the shadow the backward pass reads anyway -- written by the producer's epilogue, staged by LDS DMA
    USE_PLANES: bool = field(default_factory=_env_flag("MH_CONV_PLANES", "1"))
    # ... and the fp32 copy of such an activation is not stored when no op of the plan reads it (elision.elide_fp32_activations)
    PLANES_ONLY: bool = True
    # tests: fill every fp32 buffer whose store a plan elides with NaN when the plan is built (elision.note_elided)
    POISON_ELIDED: bool = field(default_factory=lambda: os.environ.get("MH_POISON_ELIDED", "0") == "1")
    # ... and the planes of tensors no plane kernel produces are written by THEIR producers (the level front end, the exact-fp32 layers in front of
    # conv4 / conv6, one concat-split for the context network's input) instead of by a split launch in front of every consumer
    FUSE_SPLITS: bool = True
    # ... and the INPUT GRADIENTS of those layers (and of the 1/8-resolution estimator's) run the same kernel with one plane (mh_conv2d_planes_bwd)
    PLANES_DGRAD: bool = True
    # diagnostics (bench.py --stamps): device time stamps (mh_stamp) recorded as plan ops -- the REPLAYED graph timed from the inside, without a tracer
    STAMPS: bool = False
    # input gradients stage the bf16 shadow of dz when the previous input gradient's epilogue wrote one (mh_conv2d_sh2)
    SHADOW_DGRAD: bool = True
    # ... and then do not store the fp32 gradient map at all when its only reader is such an input gradient (elision.elide_fp32_gradient_maps)
    SHADOW_ONLY: bool = True


@dataclass(frozen=True)
class DispNetSchedule(object):
    """DispNetEngine"""
    # the stride-1 3x3 layers with at least PLANES_MIN_PIX pixels run mh_conv2d_planes / mh_conv2d_planes_bwd (the K-chunked kernel beyond 128 reduction
    # channels) from bf16 planes; the coarser layers (conv5_1, conv6_1, iconv5: <= 480 pixels, 9 - 19 MB of weights for 16 - 60 workgroups) stay on the
    # split-K igemm kernels (scripts/microbench.py dispnet: 25 vs 41 us, 41 vs 73 us).  MH_CONV_PLANES=0 turns the path off.
    USE_PLANES: bool = field(default_factory=_env_flag("MH_CONV_PLANES", "1"))
    PLANES_MIN_PIX: int = 1920
    # (round 6) the stride-2 5x5 layer conv2 of both towers on the stride-2 plane kernel (conv_planes_s2fwd_kernel: split-bf16 from planes) instead of the exact-fp32
    # tiled kernel
    PLANES_S2: bool = True
    # conv3's FORWARD pass (5x5, 145 -> 256, plain bf16) on the stride-2 plane kernel too: two-row tiles (157 KB of patch) -- 2.752 / 2.756 against 2.791 / 2.792 ms (r6m)
    PLANES_S2_CONV3: bool = True
    # (round 6) the bf16 shadow of an activation / a gradient map is written by the launch that PRODUCES it -- the plane kernels' epilogue (hi plane), mh_conv2d_sh of the
    # tiled kernels; for a gradient map: its last contributor -- where a later launch reads it (plane forward layer, streamed filter gradient, plane input gradient, leaky
    # mask), instead of a shadow_cast launch in front of the reader
    PRODUCER_SHADOWS: bool = True
    # FULL momentum steps: every filter-gradient batch is followed, on its own side lane, by the momentum update of the layers it completes; the launch
    # behind the join covers what is left.  DispNet has 42 M parameters: one update over all of them is 840 MB of traffic at the very end of the step.
    EARLY_UPDATE: bool = field(default_factory=_env_flag("MH_EARLY_UPDATE", "1"))
    # the gradient buffer's zero fill on the side lane beside the forward pass
    ZERO_GRADS_EARLY: bool = field(default_factory=lambda: os.environ.get("MH_DN_ZERO_EARLY", "0") != "0")
    # filter gradients leave for a side lane in batches of FLUSH_MIN layers (one lane: 2 -> 3.15 ms, 3 -> 3.23, 4 -> 3.16, 6 -> 3.20, 12 -> 3.33),
    # the batches alternating over SIDE_LANES lanes (r04 sweep at 375x1242: 1 lane 3.21 ms, 2 lanes 3.39, 3 lanes 3.48)
    FLUSH_MIN: int = field(default_factory=lambda: int(os.environ.get("MH_DN_FLUSH_MIN", "2")))
    SIDE_LANES: int = field(default_factory=lambda: int(os.environ.get("MH_DN_LANES", "1")))
    DETERMINISTIC: bool = field(default_factory=lambda: os.environ.get("MH_DETERMINISTIC", "0") == "1")
