"""Shared-model step over RCCL on real devices.  `test_rccl_world2` spawns one process per GPU (2 ranks, RCCL over xGMI) and is skipped on a box with fewer
than two devices, so any >= 2-GPU box exercises the collective path automatically; `test_rccl_single_rank` drives the same code on a 1-rank communicator so
the 1-GPU box covers everything except the wire.  Two forms of the collective:
  in_graph = True  (round 6, the GPU default): mh_comm_init / mh_allreduce_sum behind the C-ABI, RECORDED in the step's plan -- [estimators + context + loss] on
                   a side lane where the pyramid's backward pass starts, [pyramid] behind it -- so the step is ONE captured hipGraph;
  in_graph = False (round 5): torch.distributed all-reduce(s) between two / three captured graphs (early = two pieces, else one).
Reference of both: single-process gradients of every stream summed, times 1/world, one momentum step.
`test_in_graph_collective_single_rank_equals_private_step`: on one rank the sum is the identity, so the shared-model step must reproduce the private step
BIT FOR BIT (the step has no order-dependent arithmetic since round 6) -- FULL and MAD, eager and replayed."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")
H, W = 128, 256
LR = 1e-2


def _worker(rank, world, port, q, early, in_graph):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from madnet_hip import engine as E, synthetic as S
        from madnet_hip.adapter import Adapter
        import Nets
        wn = S.calibrated_weights(dict(E.madnet_manifest()), 1)
        l, r, gt = S.make_pair(H, W, stream_id=rank)              # stream i -> rank i
        left = torch.from_numpy(l).cuda(); right = torch.from_numpy(r).cuda()
        net = Nets.get_stereo_net("MADNet", {"left_img": left, "right_img": right, "split_layers": [None], "sequence": True,
                                              "train_portion": "BEGIN", "bulkhead": False, "weights": wn, "precision": "fp32"})
        ad = Adapter(net, mode="FULL", lr=LR, shared_model=True, use_graph=True, early_reduce=early, in_graph_collective=in_graph)
        out = ad.step(l, r, gt[..., 0])
        assert (ad.comm is not None) == in_graph and len(ad._plans["FULL"]) == (1 if in_graph else (3 if early else 2))
        assert ad.collectives_last_step == (2 if (early or in_graph) else 1)
        w1 = net.engine.params.w.clone()
        g1 = net.engine.params.g.clone()
        out2 = ad.step(l, r, gt[..., 0])                            # a second replay of the captured graphs
        ws = [torch.zeros_like(w1) for _ in range(world)]
        dist.all_gather(ws, net.engine.params.w)
        same = all(bool(torch.equal(ws[0], x)) for x in ws[1:])
        q.put((rank, out["loss"], same, w1.cpu().numpy(), g1.cpu().numpy(), out2["loss"]))
    finally:
        dist.destroy_process_group()


def _reference(world):
    """one process, this GPU: gradients of stream 0 .. world-1 summed (fp32 engine)."""
    from madnet_hip import _ffi, engine as E, synthetic as S
    lib = _ffi.lib()
    wn = S.calibrated_weights(dict(E.madnet_manifest()), 1)
    gsum, lsum, w0 = None, 0.0, None
    for sid in range(world):
        l, r, gt = S.make_pair(H, W, stream_id=sid)
        eng = E.MadNetEngine(lib, H, W, B=1, device="cuda", weights=wn, precision="fp32")
        eng.set_inputs(l, r, gt[..., 0])
        eng.build_plan("FULL", lr=LR, update=False).run(lib, 0)
        torch.cuda.synchronize()
        gsum = eng.params.g.clone() if gsum is None else gsum + eng.params.g
        lsum += float(eng.res_loss[0])
        w0 = eng.params.w.clone()
    return gsum.cpu().numpy(), lsum, w0.cpu().numpy()


def _run(world, early, in_graph=False):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, early, in_graph)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    gsum, lsum, w0 = _reference(world)
    for rk in res:
        assert rk[2], "ranks diverged after the shared update"
        assert abs(rk[1] - lsum / world) <= 1e-5 * max(1.0, abs(lsum))          # the loss every rank acts on = mean over the streams
        scale = float(np.abs(gsum).max())
        assert np.abs(rk[4] - gsum).max() <= 2e-4 * scale                        # all-reduced (summed) gradient
        w_ref = w0 - LR * (gsum / world)                                          # first step: accum = g / world ; w -= lr * accum
        assert np.abs(rk[3] - w_ref).max() <= 2e-4 * LR * scale + 1e-7
        assert np.isfinite(rk[5])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (runs by itself on any >= 2-GPU box)")
@pytest.mark.parametrize("form", ["in_graph", "host_early", "host_late"])
def test_rccl_world2(hip, form):
    _run(2, form != "host_late", in_graph=(form == "in_graph"))


@pytest.mark.parametrize("form", ["in_graph", "host_early"])
def test_rccl_single_rank(hip, form):
    _run(1, True, in_graph=(form == "in_graph"))


def test_comm_abi_single_rank(hip):
    """mh_comm_* / mh_allreduce_sum through the C-ABI on a 1-rank communicator: info, identity sum of one range and of a three-range group, the same launches
    captured into a hipGraph and replayed, error paths (a pointer that is no communicator, zero ranges)."""
    import ctypes as C
    from madnet_hip import _ffi
    from madnet_hip.comm import Comm
    from madnet_hip.plan import Recorder
    lib = _ffi.lib()
    assert lib.comm_available() == 1
    comm = Comm(lib, device="cuda")
    assert comm.world == 1 and comm.rank == 0 and comm.version > 0
    x = torch.randn(1 << 20, device="cuda"); ref = x.clone()
    comm.allreduce(lib, [(x, 0, x.numel())], stream=torch.cuda.current_stream().cuda_stream)
    comm.allreduce(lib, [(x, 0, 1000), (x, 5000, 12345), (x, x.numel() - 4, 4)], stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(x, ref) and "rccl all-reduce" in lib.last_kernel().decode()
    r = Recorder()
    comm.allreduce(r, [(x, 0, 4096)])
    r.fill(C.c_void_p(x.data_ptr()), 16, 3.0, None)
    comm.allreduce(r, [(x, 0, 64), (x, 64, 64)])
    plan = r.compile()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        plan.run(lib, st.cuda_stream); st.synchronize()
        plan.capture(lib, st.cuda_stream)
        x.copy_(ref)
        plan.launch(lib, st.cuda_stream); plan.launch(lib, st.cuda_stream)
        st.synchronize()
    assert torch.equal(x[16:], ref[16:]) and bool((x[:16] == 3.0).all())
    bogus = torch.zeros(64, device="cpu")
    bufs = (C.c_void_p * 1)(x.data_ptr()); cnt = (C.c_int64 * 1)(16)
    with pytest.raises(_ffi.MadnetHipError):
        lib.allreduce_sum(bufs, cnt, 1, C.c_void_p(bogus.data_ptr()), None)
    with pytest.raises(_ffi.MadnetHipError):
        lib.allreduce_sum(bufs, cnt, 0, comm.handle, None)
    comm.close()


@pytest.mark.parametrize("mode", ["FULL", "MAD"])
def test_in_graph_collective_single_rank_equals_private_step(hip, mode):
    import json
    import Nets
    from madnet_hip import engine as E, synthetic as S
    from madnet_hip.adapter import Adapter
    wn = S.calibrated_weights(dict(E.madnet_manifest()), 1)
    cfg = json.load(open(os.path.join(PKG, "block_config", "MadNet_full.json")))
    res = []
    for shared in (False, True):
        l, r, gt = S.make_pair(H, W)
        left = torch.from_numpy(l).cuda(); right = torch.from_numpy(r).cuda()
        net = Nets.get_stereo_net("MADNet", {"left_img": left, "right_img": right, "split_layers": [None], "sequence": True, "train_portion": "BEGIN",
                                              "bulkhead": mode == "MAD", "weights": wn, "precision": "mixed"})
        ad = Adapter(net, mode=mode, block_config=cfg, lr=LR, sample_mode="SEQUENTIAL", shared_model=shared, use_graph=True, in_graph_collective=(True if shared else None))
        outs = [ad.step(l, r, gt[..., 0]) for _ in range(3)]
        if shared:
            assert ad.comm is not None and all(len(p) == 1 for p in ad._plans.values()) and ad.collectives_last_step == (2 if mode == "FULL" else 1)
        res.append((net.engine.params.w.clone(), net.engine.params.m.clone(), [o["loss"] for o in outs]))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]
