"""bench.py -- adapted stereo pairs/sec, MADNet full-backprop online adaptation, 1242x375.

python bench.py --gpus N --steps K --warmup W

N > 1 without a torch.distributed environment: bench.py re-launches itself as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` (one rank per GPU over RCCL);
under a launcher (RANK / WORLD_SIZE set, the driver's form) WORLD_SIZE must equal --gpus.

A "step" = one pass of the hot path over one synthetic KITTI-shaped pair per GPU: forward + reprojection loss + EPE/bad3 +
full backward + momentum update (the loop body of Stereo_Online_Adaptation.py:178-253), replayed as a captured hipGraph with
the inputs resident in HBM.  Streams are independent (private models): no data-path collective, weak scaling.  After W warm-up
steps the K-step region is timed --repeats times (each bracketed by barrier + synchronize, MAX over ranks); `value` is computed
from the MEDIAN region, all regions are listed under `timing`.  Rank 0 prints ONE JSON line.

Default arithmetic = 'mixed': the forward pass stays within the north-star tolerance (1e-3 px EPE vs the fp32 CPU oracle:
split-bf16 MFMA on the large 3x3 layers, exact fp32 MFMA elsewhere), gradients run on bf16 MFMA.  `paths` lists the same
step in the other arithmetic modes (exact fp32; plain bf16 = faster but OUTSIDE the tolerance) with their own EPE.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

DTYPE_LABEL = {
    "fp32": "f32 (exact fp32 MFMA)",
    "bf16": "bf16 MFMA operands, f32 accumulate, f32 storage",
    "mixed": "bf16 MFMA operands, f32 accumulate/storage (forward: split-bf16 = 3 MFMAs per product on the large 3x3 layers, exact f32 MFMA on the "
             "others, so the disparity stays within 1e-3 px; gradients: plain bf16)",
}


_OVERRIDES = []
_STDOUT_FD = None             # main(): a duplicate of the real stdout, kept for the ONE JSON line; fd 1 itself is pointed at stderr for the run
LINE_BUDGET = 6144            # bytes: the driver keeps a bounded tail of stdout; a longer line is not parsed (BENCH_r05.json: parsed null at 30.8 KB)
DETAIL_FILE = "bench_detail.json"


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _sig(x, n=6):
    """numbers of the line at n significant digits (the detail file keeps the full doubles)"""
    if isinstance(x, float):
        if x == int(x) and abs(x) < 2 ** 53:
            return int(x)                      # byte / flop / launch counts stay exact
        return float("%.*g" % (n, x))
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


_ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "mfma_issue_frac", "traffic", "algorithmic_bytes_per_step", "algorithmic_flops_per_step",
              "us_per_step", "launches_per_step", "launch_ms", "algorithmic_bytes_per_launch", "error")


def compact_line(obj):
    """The driver-facing line: the contract's keys + roofline + cpu_baseline + one number per side measurement; everything else (kernel families, launch
    tables, per-region timings, box samples, the side configurations' own rooflines) lives in the detail file the line names."""
    out = _pick(obj, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "data"))
    out["vs_baseline"] = obj.get("vs_baseline")
    if "dtype" in obj:
        out["dtype"] = obj["dtype"]
    cfg = obj.get("config", {})
    out["config"] = _pick(cfg, ("workload", "precision", "launch", "timed_region", "step_sync", "ops_per_step", "shared_model", "collectives_per_step", "concurrent_private_streams_per_gpu",
                                "final_loss", "epe_vs_synthetic_gt"))
    if isinstance(obj.get("timing"), dict):
        out["timing"] = _pick(obj["timing"], ("repeats", "ms_per_step_min", "ms_per_step_max", "timed_steps_per_repeat"))
    if isinstance(obj.get("roofline"), dict):
        out["roofline"] = _pick(obj["roofline"], _ROOF_KEYS)
    corr = {}
    for key, short in (("roofline_corr", "fwd_d5"), ("roofline_corr_bwd", "bwd_d5"), ("roofline_corr_warp_bwd", "warp_bwd_d5"), ("roofline_corr_d81_fwd", "fwd_d81"),
                       ("roofline_corr_d81_bwd", "bwd_d81")):
        r = obj.get(key)
        if isinstance(r, dict):
            corr[short] = _pick(r, ("frac", "achieved", "traffic", "algorithmic_bytes_per_launch", "launch_ms", "error"))
    if corr:
        out["roofline_corr"] = dict(corr, bound="hbm", peak=8000.0, unit="GB/s")
    for k in ("cpu_baseline", "epe_vs_oracle", "epe_tolerance", "within_tolerance", "overrides"):
        if k in obj:
            out[k] = obj[k]
    if isinstance(obj.get("step_surface"), dict):
        out["step_surface"] = _pick(obj["step_surface"], ("value", "unit", "ms_per_step", "error"))
        out["step_surface"]["what"] = "Adapter.step, new 8-bit frame pair uploaded + loss/EPE read back every step (the reference's FPS definition)"
    if isinstance(obj.get("paths"), dict):
        out["paths"] = {k: _pick(v, ("value", "ms_per_step", "epe_vs_oracle", "within_tolerance")) for k, v in obj["paths"].items() if isinstance(v, dict)}
    if isinstance(obj.get("drift"), dict):
        out["drift"] = {k: v.get("epe_vs_fp32_engine") for k, v in obj["drift"].items() if isinstance(v, dict) and "epe_vs_fp32_engine" in v}
        if "error" in obj["drift"]:
            out["drift"]["error"] = obj["drift"]["error"]
        if "deterministic_replays" in obj["drift"]:
            out["drift"]["deterministic_replays"] = obj["drift"]["deterministic_replays"]
    if isinstance(obj.get("configs"), dict):
        out["configs"] = {}
        for k, v in obj["configs"].items():
            if not isinstance(v, dict):
                continue
            e = _pick(v, ("value", "ms_per_step", "epe_vs_oracle", "within_tolerance", "error"))
            if isinstance(v.get("roofline"), dict):
                e["roofline"] = _pick(v["roofline"], ("kernel", "frac", "us_per_step"))
            if isinstance(v.get("cpu_baseline"), dict):
                e["cpu_pairs_s"] = v["cpu_baseline"].get("value")
            out["configs"][k] = e
    if isinstance(obj.get("box"), dict):
        out["box"] = _pick(obj["box"], ("replay_over_launch_sum", "healthy"))
    if isinstance(obj.get("shared_model"), dict):
        out["shared_model"] = obj["shared_model"]
    if isinstance(obj.get("rccl"), dict):
        out["rccl"] = _pick(obj["rccl"], ("world", "backend", "version", "in_graph"))
    if isinstance(obj.get("tail"), dict):
        out["tail"] = _pick(obj["tail"], ("tail_us", "side_lane_start_us"))
    if isinstance(obj.get("kernel_families"), list):
        # the three heaviest kernel families of the step, by name / time / fraction of their own roofline
        out["top_families"] = [[f.get("kernel"), f.get("us_per_step"), f.get("frac")] for f in obj["kernel_families"][:3] if isinstance(f, dict)]
    out = _sig(out)
    out["detail"] = DETAIL_FILE
    line = json.dumps(out, separators=(",", ":"))
    for k in ("top_families", "paths", "tail", "box", "timing", "drift", "shared_model", "configs", "roofline_corr"):        # (cannot happen with today's keys; never emit an unparsable line)
        if len(line) <= LINE_BUDGET:
            break
        out.pop(k, None)
        line = json.dumps(out, separators=(",", ":"))
    return line


def _emit(obj):
    """The ONE JSON line, as the LAST thing on stdout, at most LINE_BUDGET bytes (compact_line); the complete record goes to DETAIL_FILE next to
    bench.py (and to gpurun_out/ when that directory exists).  RCCL prints a version banner through C stdio when the first communicator is created --
    flush the C buffers first so it cannot trail the JSON line."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if _OVERRIDES:
        obj["overrides"] = list(_OVERRIDES)          # a --set A/B run says so in its line
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, DETAIL_FILE), "w") as f:
                    json.dump(obj, f, indent=1)
        except OSError:
            pass
    line = compact_line(obj) + "\n"
    if _STDOUT_FD is not None:
        sys.stdout.flush()
        os.write(_STDOUT_FD, line.encode())          # the process's real stdout: everything else a library prints went to stderr (main)
    else:
        sys.stdout.write(line)
        sys.stdout.flush()


def _log(msg):
    sys.stderr.write("[bench %.1fs] %s\n" % (time.perf_counter() - _T0, msg))
    sys.stderr.flush()


_T0 = time.perf_counter()


def self_spawn(argv, n):
    """`bench.py --gpus N` typed by hand: become the launcher.  Returns the children's exit code."""
    port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    _log("launching %d ranks: %s" % (n, " ".join(cmd)))
    return subprocess.call(cmd, env=env)


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


def oracle_disparity(wn, l, r):
    import torch
    from oracle import madnet as OM
    wt = {k: torch.from_numpy(v.copy()) for k, v in wn.items()}
    with torch.no_grad():
        return OM.forward(wt, torch.from_numpy(l), torch.from_numpy(r))[-1][..., 0]


class Dev(object):
    """cuda (the product) or cpu (plumbing tests of the launcher path: the emulator library through MADNET_HIP_LIB + gloo)."""

    def __init__(self, kind, local_rank):
        import torch
        self.torch, self.kind = torch, kind
        if kind == "cuda":
            self.ndev = torch.cuda.device_count()
            assert self.ndev > 0, "bench.py needs a GPU (the HIP path has no CPU fallback)"
            self.index = local_rank % self.ndev
            torch.cuda.set_device(self.index)          # before any collective: barrier()/all_reduce use the current device
            self.name = "cuda:%d" % self.index
            self.stream = torch.cuda.Stream()
            self.sh = self.stream.cuda_stream
        else:
            self.ndev, self.index, self.name, self.stream, self.sh = 0, 0, "cpu", _NoStream(), 0

    def ctx(self):
        return self.torch.cuda.stream(self.stream) if self.kind == "cuda" else _Null()

    def sync_stream(self):
        if self.kind == "cuda":
            self.stream.synchronize()

    def sync(self):
        if self.kind == "cuda":
            self.torch.cuda.synchronize()


class _NoStream(object):
    """--device cpu (plumbing tests): what the roofline helpers ask of a torch stream"""
    cuda_stream = 0

    def synchronize(self):
        pass


class _Null(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def rccl_info(dev, dist, rank, world):
    """who is in the job: world size, RCCL version, every rank's device / XCC count -- gathered once, printed by rank 0"""
    torch = dev.torch
    mine = {"rank": rank, "device": dev.name}
    ver = None
    if dev.kind == "cuda":
        pr = torch.cuda.get_device_properties(torch.cuda.current_device())
        mine.update({"name": pr.name, "arch": getattr(pr, "gcnArchName", ""), "cus": int(pr.multi_processor_count), "xccs": int(pr.multi_processor_count) // 32,
                     "hbm_gb": round(pr.total_memory / 2 ** 30, 1), "pci_bus_id": getattr(pr, "pci_bus_id", None),
                     "hip_visible": os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES"))})
        try:
            ver = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            ver = None
    ranks = [None] * world
    dist.all_gather_object(ranks, mine)
    return {"world": world, "backend": dist.get_backend(), "version": ver, "ranks": ranks}


def timed_regions(dev, dist, one_step, steps, repeats, inner=1):
    """[seconds] of `repeats` consecutive regions of inner x K steps, each bracketed by barrier + synchronize, MAX over ranks."""
    torch = dev.torch
    steps = steps * inner

    def barrier():
        dev.sync()
        if dist is not None:
            dist.barrier()
        dev.sync()

    out = []
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step()
        dev.sync_stream()
        barrier()
        out.append(time.perf_counter() - t0)
    if dist is not None:
        t = torch.tensor(out, device=dev.name, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out = [float(x) for x in t.tolist()]
    return out


def timing_block(regions, steps, inner=1):
    n = steps * inner
    ms = sorted(1e3 * x / n for x in regions)
    return {"repeats": len(ms), "ms_per_step_median": statistics.median(ms), "ms_per_step_min": ms[0], "ms_per_step_max": ms[-1],
            "ms_per_step_all": [1e3 * x / n for x in regions], "timed_steps_per_repeat": n, "inner_repetitions": inner,
            "seconds_per_repeat": [x for x in regions],
            "note": "every timed region = inner_repetitions x --steps steps, so that a region lasts >= --min-region-seconds whatever --steps says"}


def inner_reps(dev, dist, one_step, steps, min_seconds):
    """how many times the K-step block is repeated inside one timed region so that the region lasts >= min_seconds (same on every rank)"""
    if min_seconds <= 0:
        return 1
    t = timed_regions(dev, dist, one_step, steps, 1)[0]
    return max(1, int(-(-min_seconds // max(t, 1e-6))))


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
    tl, tr, tg = (torch.from_numpy(a).to(dev.name) for a in (l, r, gt[..., 0]))
    net = Nets.get_stereo_net("MADNet", {"left_img": tl, "right_img": tr, "split_layers": [None], "sequence": True,
                                         "train_portion": "BEGIN", "bulkhead": True, "weights": wn,
                                         "precision": args.precision, "warping": True, "context_net": True,
                                         "radius_d": 2, "stride": 1})
    cfg = json.load(open(os.path.join(PKG, "block_config", args.block_config)))
    # --shared-model: ONE model for all ranks -- the sampled block's gradient ranges + the loss travel as ONE all-reduce between the block's
    # backward plan and its update plan (adapter.step); SEQUENTIAL sampling keeps the ranks on the same block without exchanging the draw
    ad = Adapter(net, mode="MAD", block_config=cfg, lr=1e-4, sample_mode="SEQUENTIAL" if args.shared_model else "PROBABILITY", num_blocks=1,
                 use_graph=not args.no_graph, shared_model=args.shared_model, in_graph_collective=(False if args.host_collective else None))
    if args.rccl is not None:
        args.rccl["in_graph"] = ad.comm is not None
    for i in range(len(cfg)):
        ad._plan((i,))                       # compile + capture every block's plan outside the timed region
    last = {}

    def one_step():
        last["o"] = ad.step(tl, tr, tg)

    for _ in range(args.warmup):
        one_step()
    inner = inner_reps(dev, dist, one_step, args.steps, args.min_region_seconds)
    regions = timed_regions(dev, dist, one_step, args.steps, args.repeats, inner)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        tb = timing_block(regions, args.steps, inner)
        ms = tb["ms_per_step_median"]
        _emit({
            "metric": "adapted stereo pairs/sec (whole node), MADNet MAD modular online adaptation 1242x375",
            "value": world * 1e3 / ms, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_LABEL[args.precision], "data": "synthetic", "timing": tb,
            "config": {"workload": "MADNet MAD adaptation step via Nets.get_stereo_net + Adapter.step (host block sampling, "
                                   "forward + loss + EPE + one block's backward + update, loss read-back), %dx%d, "
                                   "1 pair/GPU/step, %s" % (W, H, args.block_config),
                       "precision": args.precision,
                       "launch": "eager plan" if args.no_graph else "hipGraph replay per sampled block",
                       "rccl": getattr(args, "rccl", None),
                       "shared_model": bool(args.shared_model), "collectives_per_step": getattr(ad, "collectives_last_step", 0) if args.shared_model else 0,
                       "fetch_counter": ad.fetch_counter, "final_loss": last["o"]["loss"], "epe_vs_synthetic_gt": last["o"]["epe"]}})


def step_surface(args, lib, dev, wn, frames=8):
    """The reference's own FPS definition (Stereo_Online_Adaptation.py:230-234,267-268): wall time of the loop INCLUDING the
    input side and the per-step host round trip.  FULL adaptation through Nets.get_stereo_net + Adapter.step, a fresh frame
    every step delivered by Data_utils.data_reader.device_prefetcher (pinned ring + copy stream: 2 x 1.4 MB of 8-bit pixels + 1.9 MB of ground truth H2D per pair), loss /
    EPE read back every step (the reset check needs them)."""
    import torch
    import Nets
    from madnet_hip import synthetic as S
    from madnet_hip.adapter import Adapter
    from Data_utils.data_reader import device_prefetcher
    H, W = args.height, args.width
    pairs = [S.make_pair(H, W, stream_id=100, frame=t) for t in range(frames)]       # frame t shifts the texture by t px (video)
    import numpy as np
    # what a camera / PNG decoder delivers: 8-bit frames (the synthetic values are integral 0..255), float32 ground truth
    pairs8 = [(l.astype(np.uint8), r.astype(np.uint8), np.ascontiguousarray(g[..., 0])) for l, r, g in pairs]
    n_steps = args.warmup + args.steps

    class Source(object):
        def __iter__(self):
            for t in range(n_steps):
                l, r, g = pairs8[t % frames]
                yield l, r, g

    z = torch.zeros(1, H, W, 3, device=dev.name)
    net = Nets.get_stereo_net("MADNet", {"left_img": z, "right_img": z, "split_layers": [None], "sequence": True,
                                         "train_portion": "BEGIN", "bulkhead": False, "weights": wn, "precision": args.precision,
                                         "warping": True, "context_net": True, "radius_d": 2, "stride": 1})
    ad = Adapter(net, mode="FULL", lr=1e-4, use_graph=not args.no_graph)
    ad._plan("FULL")
    pf = device_prefetcher(Source(), device=dev.name, depth=3, consumer_stream=ad.stream, cast=False)
    t0, k, out = None, 0, None
    for left, right, g in pf:
        if k == args.warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        out = ad.step(left, right, g)
        k += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": args.steps / dt, "unit": "pairs/s", "ms_per_step": 1e3 * dt / args.steps,
            "what": "Nets.get_stereo_net + Adapter.step(FULL), a NEW 8-bit frame pair every step through device_prefetcher (host->pinned->HBM "
                    "on a copy stream, uint8->f32 cast by Adapter.step's copy into the engine's input buffers), loss/EPE read back every step -- the reference's FPS definition (Stereo_Online_Adaptation.py:230-234,267-268)",
            "frames": frames, "final_loss": out["loss"], "resets": ad.reset_counter}


def drift_report(args, lib, dev, wn, mk, frames=8):
    """SURVEY section 7 ("multi-step drift is reported, not bounded: online SGD is chaotic"): the run's arithmetic mode and the exact-fp32 engine adapt
    side by side on the same frame-shifted synthetic video from the same weights; disparity EPE between the two after 10 and N steps, next to how
    far the fp32 disparity itself moved since step 0."""
    import torch
    from madnet_hip import synthetic as S
    H, W = args.height, args.width
    pairs = [S.make_pair(H, W, stream_id=200, frame=t) for t in range(frames)]
    ea, eb = mk(args.precision), mk("fp32")
    pa, pb = ea.build_plan("FULL", lr=1e-4), eb.build_plan("FULL", lr=1e-4)
    out = {"frames": frames, "lr": 1e-4, "what": "mean |d_%s - d_fp32| of the step's disparity after k adaptation steps (both engines start from the same weights, "
                                              "frame t = texture shifted by t px); reported, not gated" % args.precision}
    first = None
    marks = sorted(set([1, 10, args.drift_steps]))
    for k in range(1, args.drift_steps + 1):
        l, r, g = pairs[(k - 1) % frames]
        for e, p in ((ea, pa), (eb, pb)):
            e.set_inputs(l, r, g[..., 0])
            p.run(lib, 0)
        if k in marks:
            torch.cuda.synchronize()
            if first is None:
                first = eb.pred.clone()
            out["step_%d" % k] = {"epe_vs_fp32_engine": float((ea.pred - eb.pred).abs().mean().item()),
                                  "fp32_mean_abs_disparity": float(eb.pred.abs().mean().item()),
                                  "loss": float(ea.res_loss[0].item()), "loss_fp32": float(eb.res_loss[0].item())}
    wa, wb = ea.params.w, eb.params.w
    out["weights_rel_l2_after_%d" % args.drift_steps] = float(((wa - wb).norm() / wb.norm()).item())
    return out


def apply_overrides(items, lib, E, sched_cls=None):
    """--set KEY=VALUE: library hooks are applied at once; returns (per-engine attributes, Schedule fields) for every engine built: engine.NAME=v is a
    field of the engine's immutable madnet_hip.schedule.Schedule (FUSE_HEAD, TAIL_MAIN, EARLY_WGS ...), eng.NAME=v an attribute set after construction."""
    import ast
    import dataclasses
    per_engine, sched = {}, {}
    _OVERRIDES[:] = items
    # engine.NAME is checked against the schedule of the network the run builds (MADNet: Schedule, --model dispnet: DispNetSchedule): a field of the other
    # network's schedule fails loudly instead of labelling an A/B run that changed nothing
    sched_cls = sched_cls if sched_cls is not None else E.Schedule
    fields = {f.name for f in dataclasses.fields(sched_cls)}
    for it in items:
        key, _, val = it.partition("=")
        scope, _, name = key.partition(".")
        try:
            v = ast.literal_eval(val)
        except (ValueError, SyntaxError):
            v = val
        if scope == "engine":
            assert name in fields, "--set %s: no such field in madnet_hip/schedule.py: %s" % (key, sched_cls.__name__)
            sched[name] = v
        elif scope == "eng":
            per_engine[name] = v
        elif scope == "tune":
            getattr(lib, "tune_" + name)(*[int(x) for x in (v if isinstance(v, tuple) else (v,))])        # tune.conv_tile=524288,0
        elif scope == "ops":                # a module-level knob of madnet_hip/ops.py (WGRAD_STREAM_WAVES, WGRAD_STREAM_WGS): experiments
            from madnet_hip import ops as _ops
            assert hasattr(_ops, name), "--set %s: no such name in madnet_hip/ops.py" % key
            setattr(_ops, name, v)
        else:
            raise SystemExit("--set %s: scope must be engine. / eng. / tune. / ops." % key)
    return per_engine, sched


class BoxSampler(object):
    """GPU clocks / power of the bench device sampled from sysfs by a host thread every 50 ms while a timed region runs (is a slow
    box a clock / power state?): the active level of pp_dpm_sclk / pp_dpm_mclk (the line marked '*') and the hwmon sensors freq*_input (named by their
    freq*_label), power1_average / power1_input, temp1_input.  Files that do not exist are skipped; no tool is spawned.  NOTE: on some boxes of the pool
    hwmon's freq1 reads ~94 MHz whatever the load (profiles/r05_experiments.txt #10) -- every source is reported under its own name, none is "the" clock."""

    def __init__(self, index=0):
        import glob
        self.files = {}
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
        cards = [c for c in cards if os.path.exists(os.path.join(c, "pp_dpm_sclk")) or glob.glob(os.path.join(c, "hwmon/hwmon*/freq1_input"))]
        if cards:
            c = cards[min(index, len(cards) - 1)]
            for f in sorted(glob.glob(os.path.join(c, "hwmon/hwmon*/freq*_input"))):
                lab = f.replace("_input", "_label")
                try:
                    name = open(lab).read().strip()
                except Exception:
                    name = os.path.basename(f).replace("_input", "")
                self.files["hwmon_%s_mhz" % name] = (f, 1e-6)
            for key, pat, sc in (("power_w", "hwmon/hwmon*/power1_average", 1e-6), ("power_w", "hwmon/hwmon*/power1_input", 1e-6), ("temp_c", "hwmon/hwmon*/temp1_input", 1e-3)):
                g = glob.glob(os.path.join(c, pat))
                if g and key not in self.files:
                    self.files[key] = (g[0], sc)
            for nm in ("sclk", "mclk"):
                if os.path.exists(os.path.join(c, "pp_dpm_" + nm)):
                    self.files["pp_dpm_%s_mhz" % nm] = (os.path.join(c, "pp_dpm_" + nm), None)
        self.samples = []
        self._stop = None

    def _read(self):
        out = {}
        for k, (f, sc) in self.files.items():
            try:
                t = open(f).read()
                if sc is None:
                    cur = [ln for ln in t.splitlines() if ln.strip().endswith("*")]
                    if cur:
                        out[k] = float(cur[0].split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
                else:
                    out[k] = float(t.strip()) * sc
            except Exception:
                pass
        return out

    def __enter__(self):
        import threading
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                r = self._read()
                if r:
                    self.samples.append(r)
                self._stop.wait(0.05)
        self.idle = self._read()
        self._t = threading.Thread(target=loop, daemon=True)
        if self.files:
            self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self.files:
            self._t.join(timeout=1.0)
        return False

    def summary(self):
        if not self.samples:
            return None
        out = {"samples": len(self.samples), "during_replay": {}, "before": dict(self.idle)}
        for key in self.files:
            v = [x[key] for x in self.samples if key in x]
            if v:
                out["during_replay"][key] = {"min": min(v), "median": statistics.median(v), "max": max(v)}
        # the field VERDICT r04 asked for: the shader clock under the sustained replay -- the DPM level if the box exposes it, else hwmon's sclk sensor
        for key in ("pp_dpm_sclk_mhz", "hwmon_sclk_mhz"):
            if key in out["during_replay"]:
                out["sclk_mhz_during_replay"] = dict(out["during_replay"][key], source=key)
                break
        return out


def _short_regions(dev, one_step, steps, seconds=0.5, repeats=3):
    """median ms/step of `repeats` timed regions of >= `seconds` (a config of the `configs` block: shorter than the headline's five 1-s regions)"""
    inner = inner_reps(dev, None, one_step, steps, seconds)
    reg = timed_regions(dev, None, one_step, steps, repeats, inner)
    tb = timing_block(reg, steps, inner)
    return tb["ms_per_step_median"], {"repeats": repeats, "timed_steps_per_repeat": steps * inner, "ms_per_step_all": tb["ms_per_step_all"]}


def config_mad(args, lib, dev, wn, l, r, gt, BT):
    """BASELINE config 3 inside the driver's one command: MADNet MAD modular adaptation (block_config/MadNet_piramid_only.json, --sampleMode PROBABILITY
    --numBlocks 1, np.random.seed(0): SURVEY 8(d)) through Nets.get_stereo_net + Adapter.step, i.e. with the host sampling + loss read-back of the
    reference's loop body (Stereo_Online_Adaptation.py:178-253).  Returns (entry, context for the CPU leg)."""
    import numpy as np
    import torch
    import Nets
    from madnet_hip.adapter import Adapter
    tl, tr, tg = (torch.from_numpy(a).to(dev.name) for a in (l, r, gt[..., 0]))
    net = Nets.get_stereo_net("MADNet", {"left_img": tl, "right_img": tr, "split_layers": [None], "sequence": True, "train_portion": "BEGIN", "bulkhead": True,
                                         "weights": wn, "precision": args.precision, "warping": True, "context_net": True, "radius_d": 2, "stride": 1})
    cfg_name = "MadNet_piramid_only.json"
    cfg = json.load(open(os.path.join(PKG, "block_config", cfg_name)))
    np.random.seed(0)
    ad = Adapter(net, mode="MAD", block_config=cfg, lr=1e-4, sample_mode="PROBABILITY", num_blocks=1, use_graph=not args.no_graph)
    for i in range(len(cfg)):
        ad._plan((i,))
    first = ad.step(tl, tr, tg)
    dev.sync()
    pred0 = first["disparity"].detach().float().cpu().clone()           # forward pass with the PRE-update weights = the bench weights
    last = {}

    def one_step():
        last["o"] = ad.step(tl, tr, tg)

    for _ in range(args.warmup):
        one_step()
    ms, tb = _short_regions(dev, one_step, args.steps)
    tot = float(sum(ad.fetch_counter))
    shares = [c / tot for c in ad.fetch_counter]
    ent = {"metric": "adapted stereo pairs/sec, MADNet MAD modular online adaptation 1242x375", "value": 1e3 / ms, "unit": "pairs/s", "ms_per_step": ms, "timing": tb,
           "config": {"workload": "MADNet MAD adaptation step via Nets.get_stereo_net + Adapter.step (host block sampling, forward + loss + EPE + one block's backward + update, "
                                  "loss read-back every step), %dx%d, 1 pair/GPU/step, %s, sampleMode PROBABILITY, numBlocks 1, np.random.seed(0)" % (args.width, args.height, cfg_name),
                      "precision": args.precision, "launch": "hipGraph replay per sampled block", "fetch_counter": list(ad.fetch_counter),
                      "ops_per_step_by_block": [ad._plan((i,))[0].n for i in range(len(cfg))], "final_loss": last["o"]["loss"], "epe_vs_synthetic_gt": last["o"]["epe"]}}
    if not args.no_roofline:
        with dev.ctx():
            rep = BT.family_report(lib, [ad._plan((i,))[0] for i in range(len(cfg))], dev.stream, weights=shares)
        ent["roofline"] = rep["roofline"]
        ent["roofline"]["weights"] = "launch table of every block's plan, weighted by the share of the timed steps that sampled the block"
        ent["kernel_families"] = rep["kernel_families"]
        ent["kernel_time_sum_us"] = rep["kernel_time_sum_us"]
    top = max(range(len(cfg)), key=lambda i: ad.fetch_counter[i])
    return ent, {"pred0": pred0, "block_index": top, "block_vars": list(ad.blocks[top][1]), "shares": shares}


def config_dispnet(args, lib, dev, l, r, gt, BT):
    """BASELINE config 4 inside the driver's one command: DispNet (Nets/DispNet.py:75-152, block_config/dispnet_full.json is only usable with FULL: SURVEY
    App. C) full-backprop adaptation step, 81-shift correlation volume."""
    import torch
    from madnet_hip import dispnet_engine as DE, synthetic as S
    wn = S.calibrated_weights(dict(DE.dispnet_manifest()), 1)
    H, W = args.height, args.width
    mk = lambda: DE.DispNetEngine(lib, H, W, B=1, device=dev.name, weights=wn, precision=args.precision)
    e0 = mk(); e0.set_inputs(l, r, gt[..., 0])
    e0.build_plan("NONE").run(lib, 0)
    dev.sync()
    pred0 = e0.pred.cpu().clone()
    del e0
    eng = mk(); eng.set_inputs(l, r, gt[..., 0])
    plan = eng.build_plan("FULL", lr=1e-4)
    with dev.ctx():
        plan.run(lib, dev.sh)
        dev.sync_stream()
        if dev.kind == "cuda" and not args.no_graph:
            plan.capture(lib, dev.sh)

        def one_step():
            plan.launch(lib, dev.sh)
        for _ in range(args.warmup):
            one_step()
        dev.sync_stream()
        ms, tb = _short_regions(dev, one_step, args.steps)
    st = plan.stats
    flops = st.get("conv_flops", 0.0) + st.get("wgrad_flops", 0.0)
    ent = {"metric": "adapted stereo pairs/sec, DispNet full-backprop online adaptation 1242x375", "value": 1e3 / ms, "unit": "pairs/s", "ms_per_step": ms, "timing": tb,
           "config": {"workload": "DispNet FULL adaptation step (fwd + SSIM/L1 loss + EPE + bwd + momentum), %dx%d, 1 pair/GPU/step, 81-shift correlation volume" % (W, H),
                      "precision": args.precision, "launch": "hipGraph replay", "ops_per_step": plan.n, "final_loss": float(eng.res_loss[0].item()),
                      "epe_vs_synthetic_gt": float(eng.res_met[0].item()), "pred_nonzero_frac": float((eng.pred != 0).float().mean().item())},
           "step_aggregate": {"conv_gflop_per_step": flops / 1e9, "achieved_tflops": flops / (ms * 1e-3) / 1e12, "step_mfma_frac": flops / (ms * 1e-3) / 1e12 / BT.PEAK_BF16_MFMA_TFLOPS}}
    if not args.no_roofline:
        e_t = mk(); e_t.set_inputs(l, r, gt[..., 0])
        p_t = e_t.build_plan("FULL", lr=1e-4)
        with dev.ctx():
            rep = BT.family_report(lib, [p_t], dev.stream)
        ent["roofline"] = rep["roofline"]
        ent["kernel_families"] = rep["kernel_families"]
        ent["kernel_time_sum_us"] = rep["kernel_time_sum_us"]
        ent["box"] = {"replay_over_launch_sum": ms * 1e3 / rep["kernel_time_sum_us"] if rep["kernel_time_sum_us"] else None}
        del e_t, p_t
    return ent, {"pred0": pred0, "wn": wn}


def config_private(args, lib, dev, mk, S_, rank, BT, nstreams=4):
    """SURVEY 8(e) "several streams per GPU": `nstreams` independent stereo streams with PRIVATE models on this GPU, their FULL step chains as parallel
    branches of ONE hipGraph (mh_plans_run); value = pairs/s over all streams."""
    from madnet_hip.plan import MultiPlan
    H, W = args.height, args.width
    engines, plans = [], []
    for i in range(nstreams):
        e = mk(args.precision)
        e.wgrad_lanes = 0            # serial chain per branch (a fork inside a forked branch crashes hipStreamEndCapture on ROCm 7.2: profiles/r03_experiments.txt #15)
        li, ri, gi = S_.make_pair(H, W, stream_id=1000 * (rank + 1) + i)
        e.set_inputs(li, ri, gi[..., 0])
        plans.append(e.build_plan("FULL", lr=1e-4))
        engines.append(e)
    mp = MultiPlan(plans)
    with dev.ctx():
        mp.run(lib, dev.sh)
        dev.sync_stream()
        if dev.kind == "cuda":
            mp.capture(lib, dev.sh)

        def one_step():
            mp.launch(lib, dev.sh)
        for _ in range(args.warmup):
            one_step()
        dev.sync_stream()
        ms, tb = _short_regions(dev, one_step, args.steps)
    ent = {"metric": "adapted stereo pairs/sec over %d private-model streams on ONE GPU, MADNet full-backprop 1242x375" % nstreams, "value": nstreams * 1e3 / ms, "unit": "pairs/s",
           "ms_per_step": ms, "pairs_per_step": nstreams, "timing": tb,
           "config": {"workload": "MADNet FULL adaptation step x %d independent streams with private models (weights, momentum, frames), branches of one hipGraph" % nstreams,
                      "precision": args.precision, "launch": "hipGraph replay (mh_plans_run)", "ops_per_step": sum(p.n for p in plans),
                      "final_loss_per_stream": [float(e.res_loss[0].item()) for e in engines]},
           "grouped_launch_ceiling": "one grid per layer serving all streams (per-stream weight pointers, SURVEY 8(e)) computes what --streams-per-gpu %d (shared model, batched "
                                     "through the same kernels) computes plus the weight indirection: that batched form is its upper bound -- see configs.batched4" % nstreams}
    if not args.no_roofline:
        with dev.ctx():
            rep = BT.family_report(lib, [plans[0]], dev.stream)
        ent["roofline"] = rep["roofline"]
        ent["roofline"]["note"] = "launch table of ONE stream's serial chain (every launch timed alone); the %d chains share the chip in the replay" % nstreams
        ent["kernel_time_sum_us_one_stream"] = rep["kernel_time_sum_us"]
    return ent


def config_batched(args, lib, dev, wn, S_, rank, nstreams=4):
    """the same `nstreams` streams sharing ONE model, batched through the same kernels (B = nstreams): what a grouped launch with per-stream weight
    pointers would compute, minus the pointer indirection -- the ceiling of that design"""
    import numpy as np
    from madnet_hip import engine as E
    H, W = args.height, args.width
    e = E.MadNetEngine(lib, H, W, B=nstreams, device=dev.name, weights=wn, precision=args.precision)
    pairs = [S_.make_pair(H, W, stream_id=1000 * (rank + 1) + i) for i in range(nstreams)]
    e.set_inputs(np.concatenate([q[0] for q in pairs]), np.concatenate([q[1] for q in pairs]), np.concatenate([q[2][..., 0] for q in pairs]))
    plan = e.build_plan("FULL", lr=1e-4)
    with dev.ctx():
        plan.run(lib, dev.sh)
        dev.sync_stream()
        if dev.kind == "cuda":
            plan.capture(lib, dev.sh)

        def one_step():
            plan.launch(lib, dev.sh)
        for _ in range(args.warmup):
            one_step()
        dev.sync_stream()
        ms, tb = _short_regions(dev, one_step, args.steps)
    return {"metric": "adapted stereo pairs/sec over %d streams sharing one model on ONE GPU (batched), MADNet full-backprop 1242x375" % nstreams, "value": nstreams * 1e3 / ms,
            "unit": "pairs/s", "ms_per_step": ms, "pairs_per_step": nstreams, "timing": tb, "config": {"workload": "MADNet FULL adaptation step, B = %d" % nstreams, "ops_per_step": plan.n}}


def cpu_leg(fn, steps, what, cores):
    """`steps` timed calls of fn() after one warm-up call -> cpu_baseline entry (kind 'port': the torch-CPU oracle; TensorFlow cannot run here)"""
    fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "pairs/s", "cores": cores, "kind": "port", "sample": what % steps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=5, help="how many times the K-step region is timed (value = the median region)")
    ap.add_argument("--graph-copies", type=int, default=1, help="executable graphs of the step, launched in turn (experiment)")
    ap.add_argument("--step-sync", default="auto", choices=["auto", "none", "stream"],
                    help="none: the K steps of a region are enqueued back to back; stream: the host synchronises the step's stream after every step (what a consumer of the "
                         "disparity does); auto: stream for the single-stream private MADNet FULL step (measured faster), none elsewhere")
    ap.add_argument("--min-region-seconds", type=float, default=1.0,
                    help="every timed region is lengthened to at least this many seconds by repeating the K-step block inside it (0 = exactly K steps)")
    ap.add_argument("--drift-steps", type=int, default=100, help="N-step drift report: 'mixed' vs the fp32 engine after 10 and N adaptation steps on a frame-shifted synthetic video (0 = skip)")
    ap.add_argument("--mode", default="FULL", choices=["FULL", "NONE", "MAD"])
    ap.add_argument("--block-config", default="MadNet_piramid_only.json", help="MAD mode: file under block_config/")
    ap.add_argument("--model", default="madnet", choices=["madnet", "dispnet"])
    ap.add_argument("--precision", default="mixed", choices=["fp32", "bf16", "mixed"],
                    help="mixed (default) = forward inside the 1e-3 px tolerance (split-bf16 / exact fp32 MFMA), bf16 gradients; "
                         "bf16 = bf16 MFMA operands everywhere (faster, OUTSIDE the tolerance); fp32 = exact fp32 MFMA")
    ap.add_argument("--no-paths", action="store_true", help="skip the side measurements of the other arithmetic modes")
    ap.add_argument("--height", type=int, default=375)
    ap.add_argument("--width", type=int, default=1242)
    ap.add_argument("--streams-per-gpu", type=int, default=1,
                    help="B > 1: B stereo streams that SHARE one model are batched through the same kernels on each GPU "
                         "(SURVEY 8(e); loss = mean over the B pairs = synchronous data-parallel SGD); default 1 = the reference's batch-1 loop")
    ap.add_argument("--concurrent-streams", type=int, default=1,
                    help="S > 1: S INDEPENDENT stereo streams per GPU, each with its PRIVATE model / momentum / captured graph, replayed on S HIP "
                         "streams at once (SURVEY 8(e): stream i -> GPU i mod G; a batch-1 step cannot fill 256 CUs, concurrent streams can); "
                         "value = pairs/s over all streams")
    ap.add_argument("--concurrent-graphs", action="store_true",
                    help="--concurrent-streams: one hipGraph per model on a stream of its own (default: the models' step chains as parallel branches of ONE graph)")
    ap.add_argument("--early-reduce", action="store_true", help="--shared-model: force the two-piece all-reduce on a 1-rank group too (default: only when world > 1)")
    ap.add_argument("--late-reduce", action="store_true",
                    help="--shared-model: ONE all-reduce behind the whole backward pass instead of [estimators + context + loss] early / [pyramid] late")
    ap.add_argument("--host-collective", action="store_true",
                    help="--shared-model: the round-5 form -- torch.distributed all-reduce between two captured graphs -- instead of the collective recorded inside the "
                         "step's plan through the C-ABI (mh_allreduce_sum)")
    ap.add_argument("--shared-model", action="store_true",
                    help="the streams of ALL GPUs adapt ONE model: the flat gradient buffer is all-reduced (RCCL over xGMI) between "
                         "the backward plan and the momentum plan, scaled by 1/world (BASELINE config 5); default: private models, no collective")
    ap.add_argument("--wgrad-lanes", type=int, default=-1, help="side lanes for the filter gradients (default: the engine's; 0 = serial, for clean per-kernel profiles)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE", dest="overrides",
                    help="A/B override (repeatable; the JSON line lists them under 'overrides'): engine.NAME=v sets a module flag of madnet_hip/engine.py "
                         "-- a field of the engines' immutable Schedule (madnet_hip/schedule.py: FUSE_HEAD, SHADOW_ONLY, EARLY_WGS ...) --, eng.NAME=v an attribute of every engine built (fuse_front, use_bank ...), "
                         "tune.NAME=int calls the library hook mh_tune_NAME (conv_rows, conv_bank_tile, wgrad_target_pct ...)")
    ap.add_argument("--stamps", type=int, default=0, metavar="N",
                    help="also replay a STAMPED copy of the step N times (device time stamps as plan ops, engine.STAMPS) and report where the side lane starts "
                         "and how long the tail behind the last input gradient is (tail_us, side_lane_start_us) -- measured inside the untraced graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-step-surface", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` block of the default line (BASELINE configs 3 / 4 = MAD / DispNet, 4 private-model streams, their batched ceiling) and the "
                         "extra correlation rooflines")
    ap.add_argument("--extras-on-cpu", action="store_true", help="--device cpu only (tests): also walk the roofline / paths / configs code of the default line on the emulator")
    ap.add_argument("--detail", default=None, metavar="NAME", help="file name of the complete record (default bench_detail.json; written next to bench.py and into gpurun_out/)")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                    help="cpu = plumbing test of the launcher path only (emulator library via MADNET_HIP_LIB, gloo); never a result")
    args = ap.parse_args()
    if args.detail:
        global DETAIL_FILE
        DETAIL_FILE = os.path.basename(args.detail)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(sys.argv[1:], args.gpus))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks\n" % (args.gpus, world))
        sys.exit(2)
    # stdout carries the ONE JSON line and nothing else: banners of the reference-named constructors (Nets prints like the reference does), of RCCL and of
    # the other ranks go to stderr
    global _STDOUT_FD
    sys.stdout.flush()
    if rank == 0:
        _STDOUT_FD = os.dup(1)
    os.dup2(2, 1)
    dev = Dev(args.device, local_rank)
    dist = None
    if world > 1 or args.shared_model:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl" if dev.kind == "cuda" else "gloo", rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus
    rccl = rccl_info(dev, dist, rank, world) if dist is not None else None
    args.rccl = rccl

    from madnet_hip import _ffi, engine as E, dispnet_engine as DE, synthetic as S, benchtools as BT
    lib = _ffi.lib()
    eng_overrides, sched_overrides = apply_overrides(args.overrides, lib, E, DE.DispNetSchedule if args.model == "dispnet" else E.Schedule)
    H, W = args.height, args.width
    if args.mode == "MAD":
        return bench_mad(args, lib, dev, rank, world, dist)
    dispnet = args.model == "dispnet"
    shapes = dict(DE.dispnet_manifest() if dispnet else E.madnet_manifest())
    wn = S.calibrated_weights(shapes, 1)
    l, r, gt = S.make_pair(H, W, stream_id=rank)
    SB = args.streams_per_gpu
    def mk(prec, **sched_kw):
        """an engine of the run's network; sched_kw: Schedule fields on top of the --set engine.* overrides (MADNet)"""
        if dispnet:
            e = DE.DispNetEngine(lib, H, W, B=SB, device=dev.name, weights=wn, precision=prec, schedule=DE.DispNetSchedule(**dict(sched_overrides, **sched_kw)))
        else:
            e = E.MadNetEngine(lib, H, W, B=SB, device=dev.name, weights=wn, precision=prec, schedule=E.Schedule(**dict(sched_overrides, **sched_kw)))
        for k, v in eng_overrides.items():
            assert hasattr(e, k), "--set eng.%s: no such engine attribute" % k
            setattr(e, k, v)
        return e
    eng = mk(args.precision)

    def feed(e):
        if SB > 1:
            import numpy as np
            pairs = [S.make_pair(H, W, stream_id=rank * SB + i) for i in range(SB)]
            e.set_inputs(np.concatenate([q[0] for q in pairs]), np.concatenate([q[1] for q in pairs]), np.concatenate([q[2][..., 0] for q in pairs]))
        else:
            e.set_inputs(l, r, gt[..., 0])

    feed(eng)
    if args.wgrad_lanes >= 0 and hasattr(eng, "wgrad_lanes"):
        eng.wgrad_lanes = args.wgrad_lanes
    shared = args.shared_model and args.mode == "FULL"
    use_graph = (not args.no_graph) and dev.kind == "cuda"

    comm = None
    if shared and dev.kind == "cuda" and not args.host_collective and lib.comm_available() and not dispnet:
        # the collective through the C-ABI, recorded INSIDE the step's plan: one hipGraph per shared-model step (madnet_hip/comm.py); torch.distributed has
        # carried the unique id and is not touched again
        from madnet_hip.comm import Comm
        comm = Comm(lib, rank=rank, world=world, dist=dist, device=dev.name)
        if rccl is not None:
            rccl.update({"in_graph": True, "version": comm.version_string, "via": "libmadnet_hip.so: mh_comm_init / mh_allreduce_sum (MH_OP_ALLREDUCE plan op)"})
    elif rccl is not None:
        rccl["in_graph"] = False

    def make_step(e):
        """compile + validate eagerly + capture; returns (one_step, plan)"""
        if shared and comm is not None:
            plan = e.build_plan("FULL", lr=1e-4, grad_scale=1.0 / world, collective=comm)
            bare = e.build_plan("FULL", lr=1e-4, grad_scale=1.0 / world)          # the same launches without the all-reduces: collective_ms_in_step = with - without
            G, lo = e.params.g_loss, e.pyramid_range()[1]
            with dev.ctx():
                plan.run(lib, dev.sh)
                dev.sync_stream()
                if use_graph:
                    plan.capture(lib, dev.sh)
                    bare.capture(lib, dev.sh)

            def one_step():
                plan.launch(lib, dev.sh)
            one_step.n_ops = plan.n

            def collectives_only():
                comm.allreduce(lib, [(G, lo, G.numel() - lo)], stream=dev.sh)
                comm.allreduce(lib, [(G, 0, lo)], stream=dev.sh)
            one_step.collectives_only = collectives_only
            one_step.without_collectives = lambda: bare.launch(lib, dev.sh)
            one_step.pieces = [int((G.numel() - lo) * 4), int(lo * 4)]
            return one_step, plan
        if shared:
            # data-parallel SGD over all streams: grads (sum over ranks) * 1/world -> identical momentum update everywhere.  Two pieces
            # (madnet_hip/adapter.py): [estimators + context + loss] goes out asynchronously when the backward pass reaches the
            # pyramid, [pyramid] behind the pyramid's backward graph.
            early = hasattr(e, "pyramid_range") and (args.early_reduce or (world > 1 and not args.late_reduce))
            if early:
                plan, pyr = e.build_plan("FULL", lr=1e-4, grad_scale=1.0 / world, part="grad_split")
                lo = e.pyramid_range()[1]
            else:
                plan, pyr, lo = e.build_plan("FULL", lr=1e-4, grad_scale=1.0 / world, part="grad"), None, 0
            upd = e.build_plan("FULL", lr=1e-4, grad_scale=1.0 / world, part="update")
            G = e.params.g_loss if hasattr(e.params, "g_loss") else e.params.g
        else:
            plan, upd, pyr = e.build_plan(args.mode, lr=1e-4), None, None

        def reduce_and_update(launch):
            if pyr is not None:
                first = dist.all_reduce(G[lo:], async_op=True)
                launch(pyr)
                dist.all_reduce(G[:lo])
                first.wait()
            else:
                dist.all_reduce(G)                    # RCCL: torch orders it after the backward graph on the bench stream
            launch(upd)

        with dev.ctx():
            plan.run(lib, dev.sh)                   # eager once (validates every launch)
            if shared:
                reduce_and_update(lambda q: q.run(lib, dev.sh))
            dev.sync_stream()
            if use_graph:
                for q in (plan, pyr, upd):
                    if q is not None:
                        q.capture(lib, dev.sh, copies=args.graph_copies)

        def one_step():
            plan.launch(lib, dev.sh)
            if shared:
                reduce_and_update(lambda q: q.launch(lib, dev.sh))
        one_step.n_ops = plan.n + (pyr.n if pyr is not None else 0) + (upd.n if upd is not None else 0)
        if shared:
            def collectives_only():
                if pyr is not None:
                    first = dist.all_reduce(G[lo:], async_op=True)
                    dist.all_reduce(G[:lo])
                    first.wait()
                else:
                    dist.all_reduce(G)
            one_step.collectives_only = collectives_only

            def without_collectives():               # the same launches, no all-reduce: collective_ms_in_step = step with - step without
                plan.launch(lib, dev.sh)
                if pyr is not None:
                    pyr.launch(lib, dev.sh)
                upd.launch(lib, dev.sh)
            one_step.without_collectives = without_collectives
            one_step.pieces = [int((G.numel() - lo) * 4), int(lo * 4)] if pyr is not None else [int(G.numel() * 4)]
        return one_step, plan

    one_step, plan = make_step(eng)
    _log("engine built (%s), %d ops, graph=%s" % (args.precision, plan.n, use_graph))
    CS = args.concurrent_streams
    if CS > 1:
        assert dev.kind == "cuda" and not shared and use_graph, "--concurrent-streams needs hipGraph replay on a GPU, private models"
        from madnet_hip.plan import MultiPlan
        branches = not args.concurrent_graphs
        # branches of one graph: every model's chain is serial (no side lane: a fork inside a forked branch crashes hipStreamEndCapture on this
        # runtime, profiles/r03_experiments.txt #15) -- the OTHER models' chains are what fills the gaps
        lanes = 0 if branches else None
        if lanes is not None:
            eng.wgrad_lanes = lanes
            plan = eng.build_plan(args.mode, lr=1e-4)
        plans, more_engines = [plan], []           # (the engines own the buffers the captured graphs point into)
        for i in range(1, CS):
            e_i = mk(args.precision)
            if lanes is not None:
                e_i.wgrad_lanes = lanes
            li, ri, gi = S.make_pair(H, W, stream_id=1000 * (rank + 1) + i)
            e_i.set_inputs(li, ri, gi[..., 0])
            plans.append(e_i.build_plan(args.mode, lr=1e-4))
            more_engines.append(e_i)
        if args.concurrent_graphs:
            # one graph per model, each on a stream of its own (rounds 1-2: the launches do not overlap on this runtime, +3 %)
            streams = [dev.stream] + [torch.cuda.Stream() for _ in range(CS - 1)]
            for p_i, st in zip(plans[1:], streams[1:]):
                with torch.cuda.stream(st):
                    p_i.run(lib, st.cuda_stream)
                    st.synchronize()
                    p_i.capture(lib, st.cuda_stream)

            def one_step():                      # noqa: F811  -- one "step" = every stream advances by one frame
                for p_i, st in zip(plans, streams):
                    p_i.launch(lib, st.cuda_stream)
        else:
            # the S step chains as parallel branches of ONE graph (mh_plans_run)
            mp = MultiPlan(plans)
            with dev.ctx():
                mp.run(lib, dev.sh)                  # eager once (validates every launch)
                dev.sync_stream()
                mp.capture(lib, dev.sh)

            def one_step():                      # noqa: F811
                mp.launch(lib, dev.sh)
        _log("%d concurrent private streams captured (%s)" % (CS, "one graph each" if args.concurrent_graphs else "branches of one graph"))
    if args.step_sync == "auto":
        # measured per configuration (r6u, r6v; one box each, alternating): MADNet FULL private single stream -21 / -21 / -20 us and -16 / -10 us with the host wait;
        # DispNet +21, forward only +14, four private streams +570, four batched streams +13 without it -- every configuration runs the loop that serves it better
        args.step_sync = "stream" if (args.model == "madnet" and args.mode == "FULL" and CS == 1 and SB == 1 and not shared) else "none"
    if args.step_sync == "stream":
        # the host waits for every step before it launches the next (what a consumer of the disparity does; the reference's sess.run per frame).  Measured FASTER than
        # enqueueing replays back to back: the next replay's first nodes no longer contend with the previous one's filter-gradient tail (r6t: 1286.5 against 1308.3 us)
        launch_only = one_step

        def one_step():                          # noqa: F811
            launch_only()
            dev.sync_stream()
        for k in ("n_ops", "collectives_only", "without_collectives"):
            if hasattr(launch_only, k):
                setattr(one_step, k, getattr(launch_only, k))
    with dev.ctx():
        for _ in range(args.warmup):
            one_step()
        dev.sync_stream()
        inner = inner_reps(dev, dist, one_step, args.steps, args.min_region_seconds)
        with BoxSampler(dev.index) as box_sampler:
            regions = timed_regions(dev, dist, one_step, args.steps, args.repeats, inner)
    tb = timing_block(regions, args.steps, inner)
    ms = tb["ms_per_step_median"]
    _log("timed regions done: median %.3f ms/step (min %.3f, max %.3f)" % (ms, tb["ms_per_step_min"], tb["ms_per_step_max"]))

    shared_info = None
    if shared and hasattr(one_step, "collectives_only"):
        # the collective(s) alone, back to back on the bench stream (nothing to overlap with): what the step would pay if none of it hid
        # behind the pyramid's backward pass; NOTE the buffer is summed over and over here -- measured after the timed regions
        import time as _t
        with dev.ctx():
            for _ in range(3):
                one_step.collectives_only()
            dev.sync_stream()
            if dist is not None and world > 1:
                dist.barrier()
            t0 = _t.perf_counter()
            for _ in range(20):
                one_step.collectives_only()
            dev.sync_stream()
            coll_ms = (_t.perf_counter() - t0) * 1e3 / 20
        # what the collectives cost INSIDE the step: the timed step minus the same launches without the all-reduce(s)
        nocoll = timed_regions(dev, dist, one_step.without_collectives, args.steps, 1)
        ms_without = 1e3 * nocoll[0] / args.steps
        shared_info = {"collective_ms_alone": coll_ms, "collective_ms_in_step": ms - ms_without, "ms_per_step_without_collectives": ms_without,
                       "pieces_bytes": one_step.pieces,
                       "in_graph": comm is not None,
                       "order": ("[estimators + context + loss] async behind their backward pass, [pyramid] behind the pyramid's" if len(one_step.pieces) == 2
                                 else "one all-reduce behind the backward pass"),
                       "algbw_gbs": sum(one_step.pieces) / (coll_ms * 1e-3) / 1e9 if coll_ms > 0 else None}
        with dev.ctx():
            feed(eng)
            one_step()                  # leave the engine with a real step's results for the fields below
            dev.sync_stream()

    loss = float(eng.res_loss[0].item())
    epe_gt = float(eng.res_met[0].item())
    nonzero = float((eng.pred != 0).float().mean().item())
    assert nonzero >= 0.25 or dev.kind == "cpu", "degenerate synthetic network: only %.1f%% of the disparities are non-zero" % (100 * nonzero)

    st = plan.stats
    flops = st.get("conv_flops", 0.0) + st.get("wgrad_flops", 0.0)
    byts = st.get("conv_bytes", 0.0) + st.get("wgrad_bytes", 0.0)
    name = "DispNet" if dispnet else "MADNet"
    out = {
        "metric": "adapted stereo pairs/sec (whole node), %s full-backprop online adaptation 1242x375" % name,
        "value": world * SB * CS * 1e3 / ms, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_LABEL[args.precision], "data": "synthetic",
        "timing": tb,
        "config": {"workload": name + " %s adaptation step (fwd+SSIM/L1 loss+EPE+bwd+momentum), %dx%d, %d pair%s/GPU/step, %s"
                               % (args.mode, W, H, SB, "" if SB == 1 else "s",
                                  ("ONE model shared by all streams: RCCL all-reduce of the %.1f MB gradient buffer per step" % (eng.params.g.numel() * 4e-6)) if shared
                                  else ("private model per stream" if SB == 1 else "the %d streams of a GPU share one model (batched)" % SB)),
                   "precision": args.precision,
                   "concurrent_private_streams_per_gpu": CS,
                   "launch": "hipGraph replay" if use_graph else "eager plan",
                   "timed_region": "the step's launches on frames already resident in HBM (no upload, no read-back)%s; the reference's FPS definition "
                                   "(new frame + loss read-back every step) is `step_surface`" % (", the host waits for every step" if args.step_sync == "stream" else ", steps enqueued back to back"),
                   "step_sync": args.step_sync,
                   "ops_per_step": getattr(one_step, "n_ops", plan.n), "final_loss": loss, "epe_vs_synthetic_gt": epe_gt,
                   "pred_nonzero_frac": nonzero,
                   "ranks_per_device": (world + dev.ndev - 1) // dev.ndev if dev.ndev else None},
        # whole-step aggregates: algorithmic conv work of the recorded plan (every operand read / result written once) over
        # the measured step time, against the dense bf16 MFMA peak and the HBM peak (SURVEY 8(d) denominators)
        "step_aggregate": {"conv_gflop_per_step": flops / 1e9, "conv_algorithmic_mb_per_step": byts / 1e6,
                           "achieved_tflops": flops / (ms * 1e-3) / 1e12 if ms > 0 else None,
                           "step_mfma_frac": flops / (ms * 1e-3) / 1e12 / BT.PEAK_BF16_MFMA_TFLOPS if ms > 0 else None,
                           "step_hbm_frac": byts / (ms * 1e-3) / 1e9 / BT.PEAK_HBM_GBS if ms > 0 else None,
                           "conv_launches": st.get("conv_launches"), "wgrad_launches": st.get("wgrad_launches"),
                           "wgrad_ws_bytes": st.get("wgrad_ws_bytes"), "grad_bytes": st.get("grad_bytes"),
                           "wgrad_ws_over_grad": (st.get("wgrad_ws_bytes", 0.0) / st["grad_bytes"]) if st.get("grad_bytes") else None},
    }
    if shared_info is not None:
        out["shared_model"] = shared_info
    if rccl is not None:
        out["rccl"] = rccl
    if args.stamps > 0 and rank == 0 and not dispnet and not shared and CS == 1 and use_graph:
        out["tail"] = BT.tail_stamps(lib, E, mk, feed, args, dev, ms)
    if rank == 0 and world == 1 and dispnet and SB == 1 and dev.kind == "cuda" and not args.no_cpu_baseline:
        # DispNet: disparity of the run's arithmetic mode against the fp32 CPU oracle on the bench pair (the same check MADNet's line carries)
        from oracle import dispnet as OD
        wt_ = {k: torch.from_numpy(v.copy()) for k, v in wn.items()}
        with torch.no_grad():
            d_or = OD.forward(wt_, torch.from_numpy(l), torch.from_numpy(r))[-1][..., 0]
        e0 = mk(args.precision); feed(e0)
        e0.build_plan("NONE").run(lib, 0)
        torch.cuda.synchronize()
        out["epe_vs_oracle"] = float((e0.pred.cpu() - d_or).abs().mean().item())
        out["epe_tolerance"] = 1e-3
        out["within_tolerance"] = out["epe_vs_oracle"] <= 1e-3
        del e0
    extras = rank == 0 and world == 1 and not dispnet and SB == 1 and CS == 1 and not shared and (dev.kind == "cuda" or args.extras_on_cpu) and args.mode == "FULL"
    if args.extras_on_cpu:
        BT.WARMUP_LAUNCHES, BT.TABLE_REPS = 0, 1
    if extras:
        if not args.no_roofline:
            if dev.kind == "cuda":         # (the emulator's events read 0 ms)
                with dev.ctx():
                    out["roofline_fwd"], extra = BT.roofline(lib, eng, dev.stream)
                out.update(extra)
            # top-level `roofline` = the kernel family the recorded plan spends most of its time in (every op of the plan timed alone on a
            # scratch engine): algorithmic flops / time / the DENSE bf16 MFMA peak; PMC traffic keyed by kernel string
            try:
                e_t = mk(args.precision); feed(e_t)
                p_t = e_t.build_plan(args.mode, lr=1e-4)
                with dev.ctx():
                    rep = BT.family_report(lib, [p_t], dev.stream)
                out["roofline"] = rep["roofline"]
                out["kernel_families"] = rep["kernel_families"]
                tot = out["kernel_time_sum_us"] = rep["kernel_time_sum_us"]
                # Is this box throttling?  The replayed step against the sum of its launches timed ALONE in short bursts (the plan table): ~1.0 on a healthy
                # MI355X (the side lane hides about what the dependencies cost); one box of the builder's pool ran every chip-filling kernel ~1.8x slower
                # INSIDE the sustained replay while the same launches were normal stand-alone: 2.06 ms against a 1.43 ms launch sum = 1.44
                # (profiles/r04_bench_line_slow_box.json).  A reader comparing rounds or boxes should look at this number first.
                out["box"] = {"replay_over_launch_sum": (ms * 1e3) / tot if tot else None,
                              "note": "ms_per_step / sum of the plan's launches timed alone; ~1.0 = healthy, >= 1.3 = the GPU slows down under the sustained replay "
                                      "(power / thermal state of the box), not a property of the build; sclk / power: sysfs samples every 50 ms during the timed regions"}
                bs = box_sampler.summary()
                if bs:
                    out["box"].update(bs)
                # (round 5, profiles/r05_bench_line_slow_box.json: the slow box drew 436 W before the run and 662 W during the replay -- healthy ones 250 - 300 W --
                #  with the core clock reported at 2344 MHz: the number to look at beside the ratio is the power, not the clock)
                r_ = out["box"]["replay_over_launch_sum"]
                out["box"]["healthy"] = None if r_ is None else bool(r_ < 1.3)
                del e_t, p_t
            except Exception as ex:      # never let the auxiliary measurement kill the bench line
                out["roofline"] = {"error": repr(ex)}
            if not args.no_configs and dev.kind == "cuda":
                try:
                    with dev.ctx():
                        out.update(BT.corr_rooflines(lib, dev.stream, md=eng.md))
                except Exception as ex:
                    out["roofline_corr_bwd"] = {"error": repr(ex)}
            _log("roofline done")
        # every GPU measurement first, the CPU oracle (epe_vs_oracle, cpu_baseline: ~10 s of host time) last: a timeout can only cost the auxiliary keys
        preds = {}

        def pred_of(prec):
            e = mk(prec); feed(e)
            e.build_plan("NONE").run(lib, 0)
            dev.sync()
            return e.pred.cpu().clone()

        if not args.no_cpu_baseline:
            preds[args.precision] = pred_of(args.precision)
        if not args.no_paths:
            out["paths"] = {}
            for prec in ("fp32", "bf16", "mixed"):
                if prec == args.precision:
                    continue
                e2 = mk(prec); feed(e2)
                step2, _ = make_step(e2)
                with dev.ctx():
                    for _ in range(3):
                        step2()
                    dev.sync_stream()
                    reg = timed_regions(dev, None, step2, min(args.steps, 20), 3)
                ms2 = statistics.median(1e3 * x / min(args.steps, 20) for x in reg)
                out["paths"][prec] = {"dtype": DTYPE_LABEL[prec], "ms_per_step": ms2, "value": 1e3 / ms2, "unit": "pairs/s"}
                if not args.no_cpu_baseline:
                    preds[prec] = pred_of(prec)
                del e2
            _log("paths done")
        if args.drift_steps > 0 and args.precision != "fp32" and not args.no_paths:
            try:
                out["drift"] = drift_report(args, lib, dev, wn, mk)
            except Exception as ex:
                out["drift"] = {"error": repr(ex)}
            _log("drift done")
        if not args.no_step_surface:
            try:
                out["step_surface"] = step_surface(args, lib, dev, wn)
            except Exception as ex:      # never let the auxiliary measurement kill the bench line
                out["step_surface"] = {"error": repr(ex)}
            _log("step surface done")
        cfg_ctx = {}
        if not args.no_configs:
            out["configs"] = {}
            for key, fn in (("mad", lambda: config_mad(args, lib, dev, wn, l, r, gt, BT)), ("dispnet", lambda: config_dispnet(args, lib, dev, l, r, gt, BT)),
                            ("private4", lambda: (config_private(args, lib, dev, mk, S, rank, BT), None)), ("batched4", lambda: (config_batched(args, lib, dev, wn, S, rank), None))):
                try:
                    out["configs"][key], cfg_ctx[key] = fn()
                    _log("config %s: %.3f ms/step" % (key, out["configs"][key]["ms_per_step"]))
                except Exception as ex:      # never let a side configuration kill the bench line
                    if args.extras_on_cpu:
                        raise
                    out["configs"][key] = {"error": repr(ex)}
                if dev.kind == "cuda":
                    torch.cuda.empty_cache()
        if not args.no_cpu_baseline:
            d_or = oracle_disparity(wn, l, r)
            for prec, pr in preds.items():
                epe = float((pr - d_or).abs().mean().item())
                if prec == args.precision:
                    out["epe_vs_oracle"] = epe
                    out["epe_tolerance"] = 1e-3
                    out["epe_note"] = "single-step: the disparity of ONE forward pass with the bench weights against the fp32 CPU oracle (drift over consecutive adaptation steps: `drift`)"
                    out["within_tolerance"] = epe <= 1e-3
                    _log("epe_vs_oracle %.3g" % epe)
                elif "paths" in out and prec in out["paths"]:
                    out["paths"][prec]["epe_vs_oracle"] = epe
                    out["paths"][prec]["within_tolerance"] = epe <= 1e-3
            out["cpu_baseline"] = cpu_baseline(H, W, wn, l, r, gt, args.mode)
            _log("cpu baseline done")
            # the CPU legs of the side configurations (bounded: 3 + 2 oracle steps)
            cores = out["cpu_baseline"]["cores"]
            c = cfg_ctx.get("mad")
            if c:
                from oracle import madnet as OM
                e = out["configs"]["mad"]
                epe = float((c["pred0"].reshape(d_or.shape) - d_or).abs().mean().item())
                e.update({"epe_vs_oracle": epe, "epe_tolerance": 1e-3, "within_tolerance": epe <= 1e-3})
                wt = {k: torch.from_numpy(v.copy()) for k, v in wn.items()}
                acc = {k: torch.zeros_like(v) for k, v in wt.items()}
                tl_, tr_, tg_ = (torch.from_numpy(a) for a in (l, r, gt))
                e["cpu_baseline"] = cpu_leg(lambda: OM.step(wt, acc, tl_, tr_, tg_, mode="MAD", block_vars=c["block_vars"], block_index=c["block_index"], lr=1e-4), 3,
                                            "%%d MAD steps of the torch-CPU oracle (oracle/madnet.py) on the same pair, block %d (the most sampled one), torch.set_num_threads(%d)"
                                            % (c["block_index"], cores), cores)
            c = cfg_ctx.get("dispnet")
            if c:
                from oracle import dispnet as OD
                e = out["configs"]["dispnet"]
                wt = {k: torch.from_numpy(v.copy()) for k, v in c["wn"].items()}
                tl_, tr_, tg_ = (torch.from_numpy(a) for a in (l, r, gt))
                with torch.no_grad():
                    d_od = OD.forward(wt, tl_, tr_)[-1][..., 0]
                epe = float((c["pred0"].reshape(d_od.shape) - d_od).abs().mean().item())
                e.update({"epe_vs_oracle": epe, "epe_tolerance": 1e-3, "within_tolerance": epe <= 1e-3})
                acc = {k: torch.zeros_like(v) for k, v in wt.items()}
                e["cpu_baseline"] = cpu_leg(lambda: OD.step(wt, acc, tl_, tr_, tg_, mode="FULL", lr=1e-4), 2,
                                            "%%d FULL steps of the torch-CPU oracle (oracle/dispnet.py) on the same pair, torch.set_num_threads(%d)" % cores, cores)
            for key in ("private4", "batched4"):
                if isinstance(out.get("configs", {}).get(key), dict) and "value" in out["configs"][key]:
                    out["configs"][key]["cpu_baseline"] = dict(out["cpu_baseline"], note="per stream: the headline's FULL step (the streams are independent)")
                    out["configs"][key]["epe_vs_oracle"] = out.get("epe_vs_oracle")
                    out["configs"][key]["within_tolerance"] = out.get("within_tolerance")
            _log("cpu legs of the side configurations done")
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        _emit(out)


if __name__ == "__main__":
    main()
