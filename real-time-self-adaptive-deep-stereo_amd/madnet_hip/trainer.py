"""Offline training step behind the reference's Train.py surface: forward without bulkhead, multi-scale
supervised mean_l1 against the ground truth on every predicted scale (Losses/loss_factory.get_supervised_loss,
Losses/loss_factory.py:256-302), gradients of every variable, tf.train.AdamOptimizer(lr, 0.9) (Train.py:94-102) -- compiled
once into a plan of HIP kernels (madnet_hip.engine.MadNetEngine.build_plan('TRAIN')) and replayed as a hipGraph.

Data parallelism (world size > 1): every rank trains on its own batch; the flat gradient buffer is all-reduced (RCCL over
xGMI; gloo in the CPU tests) between the gradient plan and the Adam plan and scaled by 1/world, i.e. synchronous SGD on the
mean of the per-rank losses -- the multi-GPU form the reference's single-GPU script does not have."""
import numpy as np
import torch

MAX_DISP = 192.0                                  # Train.py:20


class Trainer(object):
    def __init__(self, net, lr=1e-4, loss_weights=None, loss_type="mean_l1", max_disp=MAX_DISP, use_graph=True,
                 data_parallel=False, process_group=None):
        if loss_type != "mean_l1":
            raise NotImplementedError("supervised loss '%s': the MI355X engine implements Train.py's default, mean_l1" % loss_type)
        eng = net.engine
        if not hasattr(eng, "_build_train_plan"):
            raise NotImplementedError("this engine has no offline-training plan")
        if getattr(net, "_bulkhead", False):
            print("WARNING: Train.py builds the network with bulkhead=False; this net has bulkhead=True")
        npred = len(net.get_disparities())
        # Train.py:98-99 raises when the number of weights EQUALS the number of predictions (an inverted check, SURVEY
        # App. D); the intent -- one weight per prediction -- is what is enforced here
        if loss_weights is not None and len(loss_weights) != npred:
            raise ValueError("Wrong number of loss weights provide, should provide {}".format(npred))
        self.net, self.eng, self.lib = net, eng, net._lib
        self.lr, self.loss_weights, self.max_disp = lr, loss_weights, max_disp
        self.dp, self.pg, self.world = data_parallel, process_group, 1
        if data_parallel:
            import torch.distributed as dist
            self.dist = dist
            self.world = dist.get_world_size(process_group)
        dev = eng.left.device
        self.cuda = dev.type == "cuda"
        self.use_graph = use_graph and self.cuda
        self.stream = torch.cuda.Stream(device=dev) if self.cuda else None
        self.global_step = 0
        self._plans = None
        self.npred = npred
        self._host = torch.zeros(npred * 4 + 4, pin_memory=self.cuda)

    def _build(self):
        eng = self.eng
        parts = ("grad", "update") if self.dp else ("all",)
        plans = []
        for part in parts:
            p = eng.build_plan("TRAIN", lr=self.lr, grad_scale=1.0 / self.world, part=part, loss_weights=self.loss_weights,
                               max_disp=self.max_disp)
            if self.use_graph and p.n > 0:
                with torch.cuda.stream(self.stream):
                    p.capture(self.lib, self.stream.cuda_stream)
            plans.append(p)
        self._plans = plans

    def step(self, left, right, gt):
        """left/right: [B,H,W,3], gt: [B,H,W] or [B,H,W,1] (host arrays or device tensors).  Returns the step's loss (sum over
        the scales, as Train.py fetches it), the per-scale terms, EPE / bad3 of the full-resolution prediction."""
        eng = self.eng
        if self._plans is None:
            self._build()
        sh = self.stream.cuda_stream if self.cuda else 0
        ctx = torch.cuda.stream(self.stream) if self.cuda else _null()
        with ctx:
            eng.left.copy_(_as(left, eng.left), non_blocking=True)
            eng.right.copy_(_as(right, eng.right), non_blocking=True)
            eng.gt.copy_(_as(gt, eng.gt), non_blocking=True)
            self._plans[0].launch(self.lib, sh)
            if self.dp:
                for o, c in eng.params.ranges(eng.all_vars()):
                    self.dist.all_reduce(eng.params.g[o:o + c], group=self.pg)
                self._plans[1].launch(self.lib, sh)
            n4 = 4 * self.npred
            self._host[0:n4].copy_(eng.res_loss_ms.reshape(-1), non_blocking=True)
            self._host[n4:n4 + 4].copy_(eng.res_met, non_blocking=True)
        if self.cuda:
            self.stream.synchronize()
        h = self._host.numpy()
        losses = [float(h[4 * i]) for i in range(self.npred)]
        self.global_step += 1
        return {"loss": float(np.sum(losses)), "losses": losses, "epe": float(h[4 * self.npred]), "bad3": float(h[4 * self.npred + 1]),
                "global_step": self.global_step}

    def prediction(self):
        return self.eng.pred


class _null(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _as(x, like):
    t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
    return t.to(dtype=like.dtype).reshape(like.shape)
