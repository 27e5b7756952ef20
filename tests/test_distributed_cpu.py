"""N>1 path on CPU: world_size-2 gloo processes.  (a) independent streams: rank-sharded synthetic
streams, no collective; (b) shared model: all-reduce of the flat gradient buffer between the backward
plan and the momentum plan == single-process step on the SUM of the two streams' gradients / 2.
Runs the CPU-emulated build of the kernels (tests/emul) at a tiny resolution."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")
H, W = 48, 64


def _worker(rank, world, port, q, early=True):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MH_EMUL_THREADS"] = "4"
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from madnet_hip import _ffi, engine as E, synthetic as S
        from madnet_hip.adapter import Adapter
        import Nets
        lib = _ffi.Lib(os.path.join(ROOT, "tests", "emul", "libmadnet_emul.so"))
        lib.ensure_init()
        shapes = dict(E.madnet_manifest())
        wn = S.calibrated_weights(shapes, 1)
        l, r, gt = S.make_pair(H, W, stream_id=rank)              # stream i -> rank i
        left = torch.from_numpy(l); right = torch.from_numpy(r)
        net = Nets.get_stereo_net("MADNet", {"left_img": left, "right_img": right, "split_layers": [None], "sequence": True,
                                              "train_portion": "BEGIN", "bulkhead": False, "weights": wn,
                                              "_lib": lib, "_device": "cpu"})
        ad = Adapter(net, mode="FULL", lr=1e-2, shared_model=True, use_graph=False, early_reduce=early)
        out = ad.step(l, r, gt[..., 0])
        # early: [estimators + context + loss] goes out while the pyramid's backward pass runs, [pyramid] behind it; else gradients +
        # loss travel in ONE all-reduce behind the whole backward pass
        assert ad.collectives_last_step == (2 if early else 1)
        assert len(ad._plans["FULL"]) == (3 if early else 2)
        w_after = net.engine.params.w.clone()
        g_sum = net.engine.params.g.clone()                         # all-reduced (summed) gradient
        # every rank must hold identical weights after the shared update
        ws = [torch.zeros_like(w_after) for _ in range(world)]
        dist.all_gather(ws, w_after)
        q.put((rank, out["loss"], bool(torch.equal(ws[0], ws[1])), w_after.numpy(), g_sum.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.slow
@pytest.mark.parametrize("early", [True, False])
def test_shared_model_allreduce_world2(early):
    from conftest import _emul_backend
    backend = _emul_backend()           # builds tests/emul/libmadnet_emul.so
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, early)) for r in range(2)]
    for p in procs:
        p.start()
    # reference (computed while the two ranks work): single process, gradients of both streams summed, scaled by 1/2
    from madnet_hip import engine as E, synthetic as S
    import numpy as np
    shapes = dict(E.madnet_manifest())
    wn = S.calibrated_weights(shapes, 1)
    gsum = None
    for sid in range(2):
        l, r, gt = S.make_pair(H, W, stream_id=sid)
        eng = E.MadNetEngine(backend.lib, H, W, B=1, device="cpu", weights=wn)
        eng.set_inputs(l, r, gt[..., 0])
        eng.build_plan("FULL", lr=1e-2, update=False).run(backend.lib, 0)
        gsum = eng.params.g.clone() if gsum is None else gsum + eng.params.g
        lsum = lsum + float(eng.res_loss[0]) if sid else float(eng.res_loss[0])
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][2] and res[1][2], "ranks diverged after the shared update"
    w0 = eng.params.w.clone()            # untouched (update=False)
    w_ref = w0 - 1e-2 * (0.5 * gsum)     # first step: accum = g/2 ; w -= lr*accum
    assert np.allclose(res[0][4], gsum.numpy(), rtol=1e-4, atol=1e-7 * float(gsum.abs().max()) + 1e-12)
    # the loss every rank acts on (reward / reset logic) is the mean over the streams
    assert abs(res[0][1] - res[1][1]) < 1e-9 and abs(res[0][1] - 0.5 * lsum) <= 1e-6
    assert np.allclose(res[0][3], w_ref.numpy(), rtol=1e-5, atol=1e-7)


def _mad_worker(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MH_EMUL_THREADS"] = "4"
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json
        from madnet_hip import _ffi, engine as E, synthetic as S
        from madnet_hip.adapter import Adapter
        import Nets
        lib = _ffi.Lib(os.path.join(ROOT, "tests", "emul", "libmadnet_emul.so"))
        lib.ensure_init()
        wn = S.calibrated_weights(dict(E.madnet_manifest()), 1)
        l, r, gt = S.make_pair(H, W, stream_id=rank)
        net = Nets.get_stereo_net("MADNet", {"left_img": torch.from_numpy(l), "right_img": torch.from_numpy(r), "split_layers": [None],
                                              "sequence": True, "train_portion": "BEGIN", "bulkhead": True, "weights": wn,
                                              "_lib": lib, "_device": "cpu"})
        cfg = json.load(open(os.path.join(PKG, "block_config", "MadNet_full.json")))
        # SEQUENTIAL sampling: every rank trains the same block on the same step without exchanging the draw
        ad = Adapter(net, mode="MAD", lr=1e-2, shared_model=True, use_graph=False, block_config=cfg, sample_mode="SEQUENTIAL", num_blocks=1)
        seen = []
        for _ in range(2):
            out = ad.step(l, r, gt[..., 0])
            seen.append((tuple(out["blocks"]), ad.collectives_last_step))
        w_after = net.engine.params.w.clone()
        ws = [torch.zeros_like(w_after) for _ in range(world)]
        dist.all_gather(ws, w_after)
        P = net.engine.params
        rng = P.ranges(ad._train_vars(tuple(ad.blocks_to_train)))
        q.put((rank, seen, bool(torch.equal(ws[0], ws[1])), out["loss"], rng, P.g.clone().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.slow
def test_shared_model_mad_block_is_one_collective_world2():
    """a MAD block is two ranges of the flat layout + the loss tail -- they must travel as ONE all-reduce, and the ranks must hold
    identical weights and the SUMMED block gradient afterwards."""
    from conftest import _emul_backend
    backend = _emul_backend()
    import socket
    import numpy as np
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1], "ranks trained different blocks"
    assert all(c == 1 for _, c in res[0][1]), "a MAD shared step must be ONE collective: %r" % (res[0][1],)
    assert len(res[0][4]) == 2, "a MAD block is two ranges of the flat layout"
    assert res[0][2] and res[1][2], "ranks diverged after the shared MAD update"
    assert abs(res[0][3] - res[1][3]) < 1e-9
    for o, c in res[0][4]:          # the reduced block gradient is the same buffer content on both ranks
        assert np.array_equal(res[0][5][o:o + c], res[1][5][o:o + c]) and np.abs(res[0][5][o:o + c]).max() > 0


def test_stream_sharding_is_disjoint():
    """stream i -> rank i mod G: bench.py / Adapter use make_pair(stream_id=rank)."""
    from madnet_hip import synthetic as S
    a = S.make_pair(32, 48, stream_id=0)[0]
    b = S.make_pair(32, 48, stream_id=1)[0]
    assert a.shape == b.shape and not (a == b).all()


def test_flat_layout_has_two_all_reduce_pieces():
    """madnet_manifest: the pyramid leads the flat buffers, estimators + context network follow, the loss result sits behind them --
    [pyramid] and [the rest + loss] are the two contiguous pieces of the shared-model all-reduce; a MAD block is two ranges."""
    from madnet_hip import engine as E
    P = E.Params(E.madnet_manifest(), "cpu")
    names = [n for n, _ in P.manifest]
    pyr = [n for n in names if "pyramid" in n]
    rest = [n for n in names if "pyramid" not in n]
    (o, c), = P.ranges(pyr)
    assert o == 0
    (o2, c2), = P.ranges(rest)
    assert o2 == c and o2 + c2 == P.total and P.g_loss.numel() == P.total + 4
    assert c2 > 0.7 * P.total                     # what can overlap the pyramid's backward pass
    pyr_of = {2: (1, 2, 3, 4), 3: (5, 6), 4: (7, 8), 5: (9, 10), 6: (11, 12)}          # block_config/MadNet_full.json
    for k, convs in pyr_of.items():
        bases = [E.pyr_name(i) for i in convs] + [E.est_name(k, j) for j in range(1, 7)] + ([E.ctx_name(j) for j in range(1, 8)] if k == 2 else [])
        vs = [b + sfx for b in bases for sfx in ("/weights", "/biases")]
        assert all(v in P.offset for v in vs)
        assert len(P.ranges(vs)) == 2
