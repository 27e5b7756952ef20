"""The step() surface: one iteration of the online-adaptation loop of the reference
(Stereo_Online_Adaptation.py:178-253): sample -> ONE forward with pre-update weights -> full-res
loss + EPE/bad3 -> selected backward(s) -> momentum update -> reward update -> reset check.

Every (mode, sampled blocks) combination is compiled once into a plan and replayed as a hipGraph.
Multi-GPU: streams are independent by default (private weights, no collective).  With
shared_model=True the flat GRADIENT buffer is all-reduced (RCCL over xGMI) between the backward pass
and the fused momentum update, which is exactly data-parallel SGD; the loss used for the reward /
reset decisions is averaged too so every rank samples the same block.
On the GPU the collective goes through the C-ABI (madnet_hip/comm.py: mh_comm_init, mh_allreduce_sum) and
is RECORDED in the step's plan (round 6): the shared-model step is ONE hipGraph replay.  FULL mode
issues it in two pieces: [estimators + context network + loss] (73 % of the bytes, contiguous in the
flat layout) on a side lane as soon as the backward pass reaches the pyramid, [pyramid] behind it;
MAD: the block's ranges + the loss as one RCCL group.  torch.distributed only carries the unique id.
(CPU emulator runs -- test plumbing, gloo -- keep the older form: the collective between two plans.)
"""
import collections
import os
import numpy as np
import torch

from Sampler import sampler_factory
from . import engine as E
from . import ops


def softmax(x):
    """Stereo_Online_Adaptation.py:25-27 (no max subtraction, replicated numerically)."""
    return np.exp(x) / np.sum(np.exp(x), axis=0)


class Adapter(object):
    def __init__(self, net, mode="MAD", block_config=None, lr=1e-4, momentum=0.9, sample_mode="PROBABILITY",
                 num_blocks=1, fixed_id=0, sample_frequency=1, ssim_th=0.5, reprojection_scale=1,
                 use_graph=True, shared_model=False, process_group=None, loss="reprojection", dilation=1, decay=0.99, uf=0.01,
                 optimizer="momentum", reset_optimizer=False, reward_every_step_first=False, early_reduce=None, in_graph_collective=None, fetch_inputs=None):
        """loss='proxy', dilation, decay, uf: the continual-adaptation variant (Stereo_Continual_Adaptation.py:75-112,
        205-249, 302-304): proxy-label mean_l1 loss, weight update only every `dilation`-th frame, reward update
        sample_distribution = decay * sample_distribution (+ uf * gain on the last trained blocks).
        optimizer='adam', reset_optimizer, reward_every_step_first: the live demo's variant of the loop (Demo/demo_model.py:164,
        203-208, 253-262): tf.train.AdamOptimizer(lr) instead of momentum; a reset without a weight file re-runs the initialisers,
        i.e. also clears the optimizer slots; its `first` flag is never cleared, so every step re-seeds both remembered losses."""
        if optimizer not in ("momentum", "adam"):
            raise ValueError("optimizer must be 'momentum' or 'adam'")
        self.optimizer, self.reset_optimizer, self.reward_every_step_first = optimizer, reset_optimizer, reward_every_step_first
        if mode not in ("NONE", "FULL", "MAD"):
            raise ValueError("mode must be NONE, FULL or MAD")
        if loss not in ("reprojection", "proxy"):
            raise ValueError("loss must be 'reprojection' or 'proxy'")
        self.net, self.eng, self.lib = net, net.engine, net._lib
        self.loss, self.dilation, self.decay, self.uf = loss, max(1, int(dilation)), decay, uf
        if loss == "proxy":
            if not hasattr(self.eng, "proxy"):
                raise NotImplementedError("the proxy-label loss is implemented for the MADNet engine")
            self.eng.loss_kind = "proxy"
        if reprojection_scale != 1:
            # only the MAD blocks' losses use the scaled inputs (Stereo_Online_Adaptation.py:91-107); FULL / NONE ignore the flag
            if mode == "MAD":
                if not hasattr(self.eng, "set_reprojection_scale"):
                    raise NotImplementedError("reprojectionScale != 1 is implemented for the MADNet engine")
                self.eng.set_reprojection_scale(reprojection_scale)
        self.mode, self.lr, self.momentum = mode, lr, momentum
        self.sample_frequency, self.ssim_th = sample_frequency, ssim_th
        self.shared, self.pg = shared_model, process_group
        self.world = 1
        self.comm = None               # madnet_hip.comm.Comm: the collective is recorded inside the step's plan (GPU); None: torch.distributed between two plans
        if shared_model:
            import torch.distributed as dist
            self.dist = dist
            self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # shared FULL step: all-reduce the estimator / context gradients while the pyramid's backward pass still runs (False: ONE
        # collective behind the whole backward pass).  None = when there is a wire to hide (world > 1): on a 1-rank group the second graph
        # boundary + the extra stream hand-overs cost ~0.1 ms and hide nothing (profiles/r03_experiments.txt #11).
        if early_reduce is None:
            early_reduce = self.world > 1
        self.early_reduce = bool(early_reduce) and shared_model and mode == "FULL" and hasattr(self.eng, "pyramid_range")
        dev = self.eng.left.device
        self.cuda = dev.type == "cuda"
        if shared_model and self.cuda and in_graph_collective is not False and hasattr(self.eng, "pyramid_range") and self.lib.comm_available():
            from .comm import Comm
            with torch.cuda.device(dev):
                self.comm = Comm(self.lib, rank=(self.dist.get_rank(process_group) if self.dist.is_initialized() else 0), world=self.world,
                                 dist=self.dist, group=process_group, device=dev)
        elif in_graph_collective:
            raise RuntimeError("in_graph_collective=True needs a GPU engine of MADNet and RCCL (mh_comm_available)")
        self.use_graph = use_graph and self.cuda
        self.stream = torch.cuda.Stream(device=dev) if self.cuda else None
        # frames that already sit in device memory (a prefetcher's slots) are read by the step's first node through a table the host rewrites per step
        # (ops.InputTable / mh_fetch_inputs): no copy launches in front of the captured step; host frames take the copies of _upload
        if fetch_inputs is None:
            fetch_inputs = os.environ.get("MH_FETCH_INPUTS", "1") != "0"
        self._tab = ops.InputTable(self.lib, dev) if fetch_inputs else None
        self.blocks = []
        if mode == "MAD" and not hasattr(self.eng, "record_backward") or (mode == "MAD" and net._netName != "MADNet"):
            raise NotImplementedError("MAD adaptation is only defined for MADNet (the reference's own assert "
                                      "Stereo_Online_Adaptation.py:97 fails for DispNet)")
        if mode == "MAD":
            if getattr(net, "_bulkhead", True) is False:
                print("WARNING: MAD adaptation expects the net built with bulkhead=True")
            predictions = net.get_disparities()[:-1]
            assert len(predictions) == len(block_config)           # Stereo_Online_Adaptation.py:97
            for counter, layers in enumerate(block_config):
                names = []
                for layer in layers:
                    names += [v.op_name for v in net.get_variables(layer)]
                self.blocks.append((E.LEVELS[counter], names))
            self.sampler = sampler_factory.get_sampler(sample_mode, num_blocks, fixed_id)
        self.num_actions = len(self.blocks) if mode == "MAD" else (1 if mode == "FULL" else 0)
        self.fetch_counter = [0] * self.num_actions
        self.sample_distribution = np.zeros(shape=[self.num_actions])
        self.loss_t_1 = self.loss_t_2 = 0.0
        self.last_trained_blocks = []
        self.blocks_to_train = []
        self.reset_counter = 0
        self.step_count = 0
        self._plans = {}
        self._coll_buf = None          # staging buffer of the MAD shared-model collective (step)
        self.eng.params.w0 = self.eng.params.w.clone()              # restore target (initial weights)
        self._host = torch.zeros(8, pin_memory=self.cuda)

    # -------------------------------------------------------------------------------------------
    def _plan(self, key):
        if key not in self._plans:
            eng = self.eng
            gs = 1.0 / self.world
            parts = ("grad", "update") if (self.shared and self.comm is None) else ("all",)
            if self.shared and self.comm is None and key == "FULL" and self.early_reduce:
                parts = ("grad_split", "update")
            coll = {"collective": self.comm} if (self.comm is not None and key != "NONE") else {}
            plans = []
            for part in parts:
                if self._tab is not None:
                    coll = dict(coll, inputs=self._tab)
                if key == "NONE":
                    p = eng.build_plan("NONE", part=part, inputs=self._tab)
                elif key == "FULL":
                    p = eng.build_plan("FULL", lr=self.lr, grad_scale=gs, part=part, optimizer=self.optimizer, momentum=self.momentum, **coll)
                else:
                    p = eng.build_plan("MAD", lr=self.lr, grad_scale=gs, part=part, blocks=[self.blocks[i] for i in key],
                                       optimizer=self.optimizer, momentum=self.momentum, **coll)
                for q in (p if isinstance(p, list) else [p]):
                    if self.use_graph and q.n > 0:
                        with torch.cuda.stream(self.stream):
                            q.capture(self.lib, self.stream.cuda_stream)
                    plans.append(q)
            self._plans[key] = plans
        return self._plans[key]

    def _train_vars(self, key):
        if key == "FULL":
            return self.eng.all_vars()
        if key == "NONE":
            return []
        return sum((self.blocks[i][1] for i in key), [])

    def _sample_key(self, proxy=None):
        """sample the portion(s) of the network to train (Stereo_Online_Adaptation.py:181-189) -> the plan key of this step"""
        if self.loss == "proxy" and proxy is None:
            raise ValueError("loss='proxy' needs the proxy disparity map of every frame")
        if self.mode == "MAD" and self.step_count % self.sample_frequency == 0:
            distribution = softmax(self.sample_distribution)
            self.blocks_to_train = [int(b) for b in np.asarray(self.sampler.sample(distribution)).reshape(-1)]
            for l in self.blocks_to_train:
                self.fetch_counter[l] += 1
        key = "FULL" if self.mode == "FULL" else ("NONE" if self.mode == "NONE" else tuple(self.blocks_to_train))
        if self.step_count % self.dilation != 0:        # Stereo_Continual_Adaptation.py:205: no update op on this frame
            key = "NONE"
        return key

    def _upload(self, left, right, gt=None, proxy=None):
        """frame -> the engine's input buffers: device tensors through the step's own first node (the table names them; nothing is launched here), host arrays by
        copies on the CURRENT stream"""
        eng = self.eng
        if self._tab is not None:
            dsts = [eng.left, eng.right, eng.gt] + ([eng.proxy] if hasattr(eng, "proxy") else [])
            srcs = [left, right, gt, proxy][:len(dsts)]
            if all(x is None or _direct(x, d) for x, d in zip(srcs, dsts)) and (proxy is None or len(dsts) == 4):
                self._tab.set(srcs)
                return
            self._tab.clear()
        eng.left.copy_(_as(left, eng.left), non_blocking=True)
        eng.right.copy_(_as(right, eng.right), non_blocking=True)
        if gt is not None:
            eng.gt.copy_(_as(gt, eng.gt), non_blocking=True)
        if proxy is not None:
            eng.proxy.copy_(_as(proxy, eng.proxy), non_blocking=True)

    def _readback(self):
        """results -> the pinned host buffer (on the CURRENT stream)"""
        self._host[0:4].copy_(self.eng.res_loss, non_blocking=True)
        self._host[4:8].copy_(self.eng.res_met, non_blocking=True)

    def step(self, left, right, gt=None, proxy=None):
        eng = self.eng
        key = self._sample_key(proxy)
        plans = self._plan(key)
        sh = self.stream.cuda_stream if self.cuda else 0
        ctx = torch.cuda.stream(self.stream) if self.cuda else _null()
        with ctx:
            self._upload(left, right, gt, proxy)
            plans[0].launch(self.lib, sh)
            if self.shared and self.comm is not None:
                # the collective(s) are ops of plans[0] (one hipGraph): nothing to do between plans
                self.collectives_last_step = 0 if key == "NONE" else (2 if key == "FULL" else len(key))
            elif self.shared and len(plans) == 3:
                # FULL, two pieces: plans = [forward + loss + estimator / context backward, pyramid backward, update].  The first
                # collective is asynchronous: RCCL's stream waits for plans[0], ours goes on with the pyramid.
                P = eng.params
                lo = eng.pyramid_range()[1]
                first = self.dist.all_reduce(P.g_loss[lo:P.total + 4], group=self.pg, async_op=True)
                plans[1].launch(self.lib, sh)
                self.dist.all_reduce(P.g_loss[0:lo], group=self.pg)
                first.wait()
                self.collectives_last_step = 2
                plans[2].launch(self.lib, sh)
            elif self.shared:
                # ONE collective per contiguous gradient range; the loss result sits right behind the gradient buffer
                # (engine.Params.g), so gradients + loss travel together when the last range ends there.  Sums; the 1/world factors
                # are applied by the momentum kernel (grad_scale) and on the host (loss).
                P = eng.params
                rng = P.ranges(self._train_vars(key))
                tail = (P.total, 4)
                if rng and rng[-1][0] + rng[-1][1] == P.total:
                    rng[-1] = (rng[-1][0], rng[-1][1] + 4)
                else:
                    rng.append(tail)
                if len(rng) == 1:
                    o, c = rng[0]
                    self.dist.all_reduce(P.g_loss[o:o + c], group=self.pg)
                else:
                    # MAD: the flat layout keeps the pyramid first (the FULL step's early reduction wants it contiguous), so a block is two
                    # ranges + the loss tail.  They travel as ONE collective through a staging buffer (a block is <= 1.5 M floats: the pack /
                    # unpack copies cost less than one more latency-bound all-reduce)
                    n = sum(c for _, c in rng)
                    if self._coll_buf is None or self._coll_buf.numel() < n:
                        self._coll_buf = torch.empty(n, dtype=P.g_loss.dtype, device=P.g_loss.device)
                    buf = self._coll_buf[:n]
                    torch.cat([P.g_loss[o:o + c] for o, c in rng], out=buf)
                    self.dist.all_reduce(buf, group=self.pg)
                    at = 0
                    for o, c in rng:
                        P.g_loss[o:o + c].copy_(buf[at:at + c]); at += c
                self.collectives_last_step = 1
                plans[1].launch(self.lib, sh)
            self._readback()
        if self.cuda:
            self.stream.synchronize()
        return self._finish()

    def _finish(self):
        """host side of a step whose results sit in the pinned buffer: reward update, reset check, bookkeeping"""
        eng = self.eng
        new_loss = float(self._host[0]) / (self.world if self.shared else 1)
        epe = float(self._host[4]); bad3 = float(self._host[5])
        # ---- reward update of the sampling logits (Stereo_Online_Adaptation.py:211-224)
        if self.mode == "MAD":
            if self.step_count == 0 or self.reward_every_step_first:
                self.loss_t_2 = new_loss
                self.loss_t_1 = new_loss
            expected_loss = 2 * self.loss_t_1 - self.loss_t_2
            gain_loss = expected_loss - new_loss
            self.sample_distribution = self.decay * self.sample_distribution
            for i in self.last_trained_blocks:
                self.sample_distribution[i] += self.uf * gain_loss
            self.last_trained_blocks = self.blocks_to_train
            self.loss_t_2 = self.loss_t_1
            self.loss_t_1 = new_loss
        # ---- reset to the initial weights if the loss explodes (:241-244); momentum is NOT reset (the demo without a weight file
        # re-initialises everything, optimizer slots included: reset_optimizer)
        did_reset = False
        if new_loss > self.ssim_th:
            with (torch.cuda.stream(self.stream) if self.cuda else _null()):      # same stream as the next step's plan
                eng.params.w.copy_(eng.params.w0)
                if self.reset_optimizer:
                    eng.params.m.zero_()
                    if getattr(eng.params, "v", None) is not None:
                        eng.params.v.zero_()
                    if getattr(eng, "adam_state", None) is not None:
                        eng.adam_state.copy_(torch.tensor([0.9, 0.999]))
            self.reset_counter += 1
            did_reset = True
        self.step_count += 1
        return {"epe": epe, "bad3": bad3, "loss": new_loss, "disparity": eng.pred,
                "blocks": list(self.blocks_to_train) if self.mode == "MAD" else [], "reset": did_reset}


class MultiAdapter(object):
    """S stereo streams with PRIVATE models on ONE GPU ("several streams per GPU may be batched"): every stream keeps its own
    Adapter -- weights, optimizer state, sampler, reward logic, reset -- and one step() advances all of them; the S step plans run as parallel
    branches of ONE hipGraph (mh_plans_run), one graph per combination of plan keys (FULL / NONE: one; MAD: the sampled blocks of every stream).
    The models' chains are serial inside their branch (engine.wgrad_lanes = 0, set here before any plan is built)."""

    def __init__(self, adapters):
        from .plan import MultiPlan
        self._MultiPlan = MultiPlan
        self.adapters = list(adapters)
        assert self.adapters and all(not a.shared for a in self.adapters), "private models only"
        a0 = self.adapters[0]
        self.lib, self.cuda = a0.lib, a0.cuda
        for a in self.adapters:
            assert not a._plans, "build the MultiAdapter before the adapters' first step"
            a.eng.wgrad_lanes = 0
            a.use_graph = False            # the combination is captured here, not the single plans
        self.stream = a0.stream
        for a in self.adapters:
            a.stream = self.stream         # ONE stream orders uploads, the graph, read-backs and resets of every model
        self._graphs = collections.OrderedDict()       # plan-key tuple -> captured MultiPlan, least recently used first

    # MAD with S streams can sample 5^S block combinations: the captured graphs are kept in an LRU of this many entries (evicted graph execs are
    # destroyed; a combination that comes back is captured again, ~1 ms)
    MAX_GRAPHS = 64

    def step(self, frames):
        """frames: one (left, right[, gt[, proxy]]) tuple per stream -> [Adapter.step()'s dict per stream]"""
        assert len(frames) == len(self.adapters)
        for a, f in zip(self.adapters, frames):          # every frame is checked BEFORE any sampler advances
            if a.loss == "proxy" and (len(f) <= 3 or f[3] is None):
                raise ValueError("loss='proxy' needs the proxy disparity map of every frame")
        keys = tuple(a._sample_key(f[3] if len(f) > 3 else None) for a, f in zip(self.adapters, frames))
        mp = self._graphs.get(keys)
        if mp is not None:
            self._graphs.move_to_end(keys)
        sh = self.stream.cuda_stream if self.cuda else 0
        ctx = torch.cuda.stream(self.stream) if self.cuda else _null()
        with ctx:
            for a, f in zip(self.adapters, frames):
                a._upload(*f)
            if mp is None:
                mp = self._MultiPlan([a._plan(k)[0] for a, k in zip(self.adapters, keys)])
                if self.cuda:
                    mp.capture(self.lib, sh)
                self._graphs[keys] = mp
                while len(self._graphs) > self.MAX_GRAPHS:
                    _, old = self._graphs.popitem(last=False)
                    if old.graph is not None:
                        self.stream.synchronize()            # its last replay may still be in flight
                        self.lib.graph_destroy(old.graph); old.graph = None
            mp.launch(self.lib, sh)
            for a in self.adapters:
                a._readback()
        if self.cuda:
            self.stream.synchronize()
        return [a._finish() for a in self.adapters]


class _null(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _direct(x, like):
    """can the step's first node read frame x itself?  a device tensor of the buffer's size, uint8 or float32, contiguous, 16-byte aligned"""
    return (torch.is_tensor(x) and x.device == like.device and x.dtype in (torch.uint8, torch.float32) and x.is_contiguous() and x.numel() == like.numel()
            and x.data_ptr() % 16 == 0)


def _as(x, like):
    if torch.is_tensor(x) and x.device == like.device:
        return x.reshape(like.shape)            # (a uint8 frame from device_prefetcher(cast=False): copy_ casts while it copies)
    t = torch.as_tensor(x, dtype=torch.float32)
    return t.reshape(like.shape)
