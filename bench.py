"""bench.py -- adapted stereo pairs/sec, MADNet full-backprop online adaptation, 1242x375.

python bench.py --gpus N --steps K --warmup W   (N>1: launched by torch.distributed.run, one rank
per GPU).  A "step" = one pass of the hot path over one synthetic KITTI-shaped pair per GPU:
forward + reprojection loss + EPE/bad3 + full backward + momentum update (the loop body of
Stereo_Online_Adaptation.py:178-253), replayed as a captured hipGraph.  Streams are independent
(private models): no data-path collective, weak scaling.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def _emit(obj):
    """The ONE JSON line, as the LAST thing on stdout: RCCL prints a version banner through C stdio when the first
    communicator is created -- flush the C buffers first so it cannot trail the JSON line."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.write(json.dumps(obj) + "\n")
    sys.stdout.flush()


def _log(msg):
    sys.stderr.write("[bench %.1fs] %s\n" % (time.perf_counter() - _T0, msg))
    sys.stderr.flush()


_T0 = time.perf_counter()


def cpu_baseline(H, W, wn, l, r, gt, mode, steps=8):
    """CPU stand-in for the reference TF1 CPU path (TensorFlow cannot run here): the torch-CPU
    oracle executing the identical step on the same inputs and weights, on the host cores of this
    box.  Bounded sample: 2 warm-up + `steps` timed steps (~10-30 s)."""
    import torch
    from oracle import madnet as OM
    cores = min(os.cpu_count() or 1, 64)     # oneDNN stops scaling (and oversubscribes) far below 256 threads
    torch.set_num_threads(cores)
    wt = {k: torch.from_numpy(v.copy()) for k, v in wn.items()}
    acc = {k: torch.zeros_like(v) for k, v in wt.items()}
    tl, tr, tg = (torch.from_numpy(a) for a in (l, r, gt))
    for _ in range(2):
        OM.step(wt, acc, tl, tr, tg, mode=mode, lr=1e-4)
    t0 = time.perf_counter()
    for _ in range(steps):
        OM.step(wt, acc, tl, tr, tg, mode=mode, lr=1e-4)
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "%d %s steps of the torch-CPU oracle (oracle/madnet.py) on the same %dx%d pair, "
                      "torch.set_num_threads(%d)" % (steps, mode, W, H, cores)}


def epe_vs_oracle(lib, H, W, wn, l, r, gt, precision="fp32"):
    """mean |d_hip - d_oracle| of disparities[-1] on identical inputs and weights (single forward)."""
    import torch
    from madnet_hip import engine as E
    from oracle import madnet as OM
    eng = E.MadNetEngine(lib, H, W, B=1, device="cuda", weights=wn, precision=precision)
    eng.set_inputs(l, r, gt[..., 0])
    eng.build_plan("NONE").run(lib, 0)
    torch.cuda.synchronize()
    wt = {k: torch.from_numpy(v.copy()) for k, v in wn.items()}
    with torch.no_grad():
        d = OM.forward(wt, torch.from_numpy(l), torch.from_numpy(r))[-1][..., 0]
    return float((eng.pred.cpu() - d).abs().mean().item())


def bench_mad(args, lib, dev, rank, world, dist):
    """BASELINE config 3: MAD modular adaptation through the reference's own API surface (Nets.get_stereo_net +
    Adapter.step): per step the host samples a block (Sampler/sampler_factory.py), replays that block's captured
    graph (forward + full-res loss/metrics + block backward + block update), reads loss/EPE back and updates the
    sampling logits -- i.e. a step includes the host round trip, exactly like the reference loop body."""
    import torch
    import Nets
    from madnet_hip import engine as E, synthetic as S
    from madnet_hip.adapter import Adapter
    H, W = args.height, args.width
    wn = S.calibrated_weights(dict(E.madnet_manifest()), 1)
    l, r, gt = S.make_pair(H, W, stream_id=rank)
    tl, tr, tg = (torch.from_numpy(a).to(dev) for a in (l, r, gt[..., 0]))
    net = Nets.get_stereo_net("MADNet", {"left_img": tl, "right_img": tr, "split_layers": [None], "sequence": True,
                                         "train_portion": "BEGIN", "bulkhead": True, "weights": wn,
                                         "precision": args.precision, "warping": True, "context_net": True,
                                         "radius_d": 2, "stride": 1})
    cfg = json.load(open(os.path.join(PKG, "block_config", args.block_config)))
    ad = Adapter(net, mode="MAD", block_config=cfg, lr=1e-4, sample_mode="PROBABILITY", num_blocks=1,
                 use_graph=not args.no_graph)
    for i in range(len(cfg)):
        ad._plan((i,))                       # compile + capture every block's plan outside the timed region
    for _ in range(args.warmup):
        ad.step(tl, tr, tg)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_step = ad.step(tl, tr, tg)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        _emit({
            "metric": "adapted stereo pairs/sec (whole node), MADNet MAD modular online adaptation 1242x375",
            "value": world * args.steps / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": "MADNet MAD adaptation step via Nets.get_stereo_net + Adapter.step (host block sampling, "
                                   "forward + loss + EPE + one block's backward + update, loss read-back), %dx%d, "
                                   "1 pair/GPU/step, %s" % (W, H, args.block_config),
                       "launch": "eager plan" if args.no_graph else "hipGraph replay per sampled block",
                       "fetch_counter": ad.fetch_counter, "final_loss": out_step["loss"], "epe_vs_synthetic_gt": out_step["epe"]}})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", default="FULL", choices=["FULL", "NONE", "MAD"])
    ap.add_argument("--block-config", default="MadNet_piramid_only.json", help="MAD mode: file under block_config/")
    ap.add_argument("--model", default="madnet", choices=["madnet", "dispnet"])
    ap.add_argument("--precision", default="bf16", choices=["fp32", "bf16"],
                    help="bf16 (default, BASELINE.json's config) = bf16 MFMA inputs, fp32 accumulate/storage; "
                         "fp32 = exact fp32 MFMA (the parity path, also timed and reported as `parity_path`)")
    ap.add_argument("--no-parity-path", action="store_true", help="skip the fp32 side measurement of a bf16 run")
    ap.add_argument("--height", type=int, default=375)
    ap.add_argument("--width", type=int, default=1242)
    ap.add_argument("--streams-per-gpu", type=int, default=1,
                    help="B > 1: B stereo streams that SHARE one model are batched through the same kernels on each GPU "
                         "(SURVEY 8(e); loss = mean over the B pairs = synchronous data-parallel SGD); default 1 = the reference's batch-1 loop")
    ap.add_argument("--shared-model", action="store_true",
                    help="the streams of ALL GPUs adapt ONE model: the flat gradient buffer is all-reduced (RCCL over xGMI) between "
                         "the backward plan and the momentum plan, scaled by 1/world (BASELINE config 5); default: private models, no collective")
    ap.add_argument("--wgrad-lanes", type=int, default=-1, help="side lanes for the filter gradients (default: the engine's; 0 = serial, for clean per-kernel profiles)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank != 0:
        os.dup2(2, 1)                      # only rank 0 owns stdout (the ONE JSON line); library banners of the others -> stderr
    torch.cuda.set_device(local_rank)          # before any collective: barrier()/all_reduce use the current device
    dev = "cuda:%d" % local_rank
    dist = None
    if world > 1 or args.shared_model:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from madnet_hip import _ffi, engine as E, dispnet_engine as DE, synthetic as S, benchtools as BT
    lib = _ffi.lib()
    H, W = args.height, args.width
    if args.mode == "MAD":
        return bench_mad(args, lib, dev, rank, world, dist)
    dispnet = args.model == "dispnet"
    shapes = dict(DE.dispnet_manifest() if dispnet else E.madnet_manifest())
    wn = S.calibrated_weights(shapes, 1)
    l, r, gt = S.make_pair(H, W, stream_id=rank)
    SB = args.streams_per_gpu
    eng = (DE.DispNetEngine(lib, H, W, B=SB, device=dev, weights=wn, precision=args.precision) if dispnet
           else E.MadNetEngine(lib, H, W, B=SB, device=dev, weights=wn, precision=args.precision))
    if SB > 1:
        import numpy as np
        pairs = [S.make_pair(H, W, stream_id=rank * SB + i) for i in range(SB)]
        eng.set_inputs(np.concatenate([q[0] for q in pairs]), np.concatenate([q[1] for q in pairs]), np.concatenate([q[2][..., 0] for q in pairs]))
    else:
        eng.set_inputs(l, r, gt[..., 0])
    if args.wgrad_lanes >= 0 and hasattr(eng, "wgrad_lanes"):
        eng.wgrad_lanes = args.wgrad_lanes
    shared = args.shared_model and args.mode == "FULL"
    if shared:
        # data-parallel SGD over all streams: grads (sum over ranks) * 1/world -> identical momentum update everywhere
        plan = eng.build_plan("FULL", lr=1e-4, grad_scale=1.0 / world, part="grad")
        plan_upd = eng.build_plan("FULL", lr=1e-4, grad_scale=1.0 / world, part="update")
    else:
        plan = eng.build_plan(args.mode, lr=1e-4)
        plan_upd = None
    stream = torch.cuda.Stream()
    sh = stream.cuda_stream
    _log("engine built, %d ops" % plan.n)

    def one_step():
        plan.launch(lib, sh)
        if shared:
            dist.all_reduce(eng.params.g)       # RCCL on the bench stream (torch orders it after the backward graph)
            plan_upd.launch(lib, sh)

    with torch.cuda.stream(stream):
        plan.run(lib, sh)                       # eager once (validates every launch)
        if shared:
            dist.all_reduce(eng.params.g)
            plan_upd.run(lib, sh)
        stream.synchronize()
        _log("eager step ok")
        if not args.no_graph:
            plan.capture(lib, sh)
            if shared:
                plan_upd.capture(lib, sh)
            _log("hipGraph captured")
        for _ in range(args.warmup):
            one_step()
        stream.synchronize()

        def barrier():
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        stream.synchronize()
        barrier()
        dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    def timed_replay(e, steps, warm=3):
        """ms/step of a second engine's FULL plan (hipGraph replay) -- side measurement, rank 0 / N=1 only."""
        p2 = e.build_plan(args.mode, lr=1e-4)
        with torch.cuda.stream(stream):
            p2.run(lib, sh); stream.synchronize()
            if not args.no_graph:
                p2.capture(lib, sh)
            for _ in range(warm):
                p2.launch(lib, sh)
            stream.synchronize()
            t = time.perf_counter()
            for _ in range(steps):
                p2.launch(lib, sh)
            stream.synchronize()
            return 1e3 * (time.perf_counter() - t) / steps

    loss = float(eng.res_loss[0].item())
    epe_gt = float(eng.res_met[0].item())
    nonzero = float((eng.pred != 0).float().mean().item())
    assert nonzero >= 0.25, "degenerate synthetic network: only %.1f%% of the disparities are non-zero" % (100 * nonzero)

    out = {
        "metric": "adapted stereo pairs/sec (whole node), %s full-backprop online adaptation 1242x375" % ("DispNet" if dispnet else "MADNet"),
        "value": world * SB * args.steps / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
        "config": {"workload": ("DispNet" if dispnet else "MADNet") + " %s adaptation step (fwd+SSIM/L1 loss+EPE+bwd+momentum), %dx%d, %d pair%s/GPU/step, %s"
                               % (args.mode, W, H, SB, "" if SB == 1 else "s",
                                  ("ONE model shared by all streams: RCCL all-reduce of the %.1f MB gradient buffer per step" % (eng.params.g.numel() * 4e-6)) if shared
                                  else ("private model per stream" if SB == 1 else "the %d streams of a GPU share one model (batched)" % SB)),
                   "launch": "eager plan" if args.no_graph else "hipGraph replay",
                   "ops_per_step": plan.n, "final_loss": loss, "epe_vs_synthetic_gt": epe_gt,
                   "pred_nonzero_frac": nonzero},
    }
    _log("timed region done: %.3f ms/step" % (1e3 * dt / args.steps))
    if rank == 0 and world == 1 and not dispnet and SB == 1 and not shared:
        if not args.no_roofline:
            with torch.cuda.stream(stream):
                out["roofline"], extra = BT.roofline(lib, eng, stream)
            out.update(extra)
            _log("roofline done")
        if args.precision != "fp32" and not args.no_parity_path:
            e32 = E.MadNetEngine(lib, H, W, B=1, device=dev, weights=wn, precision="fp32")
            e32.set_inputs(l, r, gt[..., 0])
            ms32 = timed_replay(e32, min(args.steps, 20))
            out["parity_path"] = {"dtype": "f32", "ms_per_step": ms32, "value": 1e3 / ms32, "unit": "pairs/s",
                                  "note": "same step with exact fp32 MFMA (the arithmetic the parity tests pin to the oracle)"}
            del e32
            _log("parity path done")
        if not args.no_cpu_baseline:
            out["epe_vs_oracle"] = epe_vs_oracle(lib, H, W, wn, l, r, gt, args.precision)
            if "parity_path" in out:
                out["parity_path"]["epe_vs_oracle"] = epe_vs_oracle(lib, H, W, wn, l, r, gt, "fp32")
            _log("epe_vs_oracle done")
            out["cpu_baseline"] = cpu_baseline(H, W, wn, l, r, gt, args.mode)
            _log("cpu baseline done")
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        _emit(out)


if __name__ == "__main__":
    main()
