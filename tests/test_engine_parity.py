"""End-to-end parity of the MADNet engine (forward + FULL / MAD adaptation step) against the
torch oracle: disparity EPE <= 1e-3 (north-star tolerance), gradients and post-step weights."""
import json
import os

import numpy as np
import pytest
import torch

from madnet_hip import engine as E
from madnet_hip import synthetic as S
from oracle import madnet as OM

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "real-time-self-adaptive-deep-stereo_amd")
EPE_TOL = 1e-3          # BASELINE.json north_star: "within 1e-3 EPE of the reference"


def _setup(backend, H, W, seed=1):
    shapes = OM.variable_shapes()
    wn = S.calibrated_weights(shapes, seed)
    l, r, gt = S.make_pair(H, W)
    eng = E.MadNetEngine(backend.lib, H, W, B=1, device=backend.device, weights=wn)
    eng.set_inputs(l, r, gt[..., 0])
    wt = {k: torch.from_numpy(v.copy()) for k, v in wn.items()}
    acc = {k: torch.zeros_like(v) for k, v in wt.items()}
    return eng, wn, wt, acc, tuple(torch.from_numpy(a) for a in (l, r, gt))


def _check(eng, wn, wt, o, backend, gtol=2e-3):
    backend.sync()
    d_ref = o["disparity"][..., 0]
    pred = eng.pred.cpu()
    epe = (pred - d_ref).abs().mean().item()
    assert epe <= EPE_TOL, "disparity EPE vs oracle %g" % epe
    assert abs(eng.res_loss[0].item() - o["loss"]) <= 2e-5 * max(1.0, abs(o["loss"]))
    assert abs(eng.res_met[0].item() - o["epe"]) <= 1e-4 * max(1.0, o["epe"])
    assert abs(eng.res_met[1].item() - o["bad3"]) <= 1e-4
    gmax = max(g.abs().max().item() for g in o["grads"].values()) if o["grads"] else 1.0
    for n, g in o["grads"].items():
        ge = eng.params.tensor(n, "g").cpu()
        # fp32 sums of ~1e5 signed terms in a different order: judge each tensor by its relative L2
        # error, and its worst element against the tensor's own scale
        rel = (ge - g).norm().item() / max(g.norm().item(), 1e-30)
        err = (ge - g).abs().max().item()
        assert rel <= gtol or g.abs().max().item() <= 1e-6 * gmax, (n, rel)
        assert err <= 10 * gtol * max(g.abs().max().item(), 1e-6 * gmax), (n, err, g.abs().max().item())
    for n in wt:
        we = eng.params.tensor(n).cpu()
        if n in o["grads"]:
            assert (we - wt[n]).abs().max().item() <= 1e-5 * max(1.0, wt[n].abs().max().item()), n
        else:   # untouched variables must be BIT-identical (MAD trains only the block)
            assert torch.equal(we, torch.from_numpy(wn[n])), n
    return epe


SIZES = [pytest.param("emul", (60, 100), id="emul-60x100"),
         pytest.param("hip", (60, 100), marks=pytest.mark.gpu, id="hip-60x100"),
         pytest.param("hip", (375, 1242), marks=pytest.mark.gpu, id="hip-375x1242")]


def _backend(name):
    from conftest import _emul_backend, _hip_backend
    return _emul_backend() if name == "emul" else _hip_backend()


@pytest.mark.parametrize("bname,size", SIZES)
def test_full_step(bname, size):
    backend = _backend(bname)
    eng, wn, wt, acc, (l, r, gt) = _setup(backend, *size)
    lr = 1e-2
    plan = eng.build_plan("FULL", lr=lr)
    plan.run(backend.lib, 0)
    o = OM.step(wt, acc, l, r, gt, mode="FULL", lr=lr)
    _check(eng, wn, wt, o, backend)


@pytest.mark.parametrize("bname,size", SIZES[:2] + [pytest.param("hip", (375, 1242), marks=pytest.mark.gpu, id="hip-375x1242")])
@pytest.mark.parametrize("cfg,block", [("MadNet_full.json", 0), ("MadNet_full.json", 3), ("MadNet_full.json", 4),
                                       ("MadNet_piramid_only.json", 2), ("MadNet_piramid_only.json", 4)])
def test_mad_step(bname, size, cfg, block):
    backend = _backend(bname)
    if bname == "emul" and block not in (0, 4):
        pytest.skip("CPU emulator: only the coarsest and the finest block (time)")
    if bname == "emul" and cfg != "MadNet_full.json":
        pytest.skip("CPU emulator: one block config (time)")
    eng, wn, wt, acc, (l, r, gt) = _setup(backend, *size)
    blocks = json.load(open(os.path.join(PKG, "block_config", cfg)))
    lv = OM.layer_variables()
    bv = sum([lv[n] for n in blocks[block]], [])
    level = E.LEVELS[block]
    lr = 1e-2
    plan = eng.build_plan("MAD", lr=lr, block_vars=bv, block_level=level)
    plan.run(backend.lib, 0)
    o = OM.step(wt, acc, l, r, gt, mode="MAD", block_vars=bv, block_index=block, lr=lr)
    _check(eng, wn, wt, o, backend)


@pytest.mark.parametrize("bname,size", SIZES[:2])
def test_full_step_without_warping(bname, size):
    """warping=False (MadNet.py:282-285,301-304,320-323,339-342): the right features enter the cost volumes un-warped; the
    upsampled disparity still feeds the estimators, so its gradient path (resize) stays."""
    backend = _backend(bname)
    shapes = OM.variable_shapes()
    wn = S.calibrated_weights(shapes, 1)
    l, r, gt = S.make_pair(*size)
    eng = E.MadNetEngine(backend.lib, size[0], size[1], B=1, device=backend.device, weights=wn, warping=False)
    eng.set_inputs(l, r, gt[..., 0])
    wt = {k: torch.from_numpy(v.copy()) for k, v in wn.items()}
    acc = {k: torch.zeros_like(v) for k, v in wt.items()}
    lr = 1e-2
    eng.build_plan("FULL", lr=lr).run(backend.lib, 0)
    o = OM.step(wt, acc, torch.from_numpy(l), torch.from_numpy(r), torch.from_numpy(gt), mode="FULL", lr=lr, warping=False)
    _check(eng, wn, wt, o, backend)
    # and it IS a different network: the warped forward gives another disparity
    with torch.no_grad():
        dw = OM.forward({k: torch.from_numpy(v) for k, v in wn.items()}, torch.from_numpy(l), torch.from_numpy(r))[-1][..., 0]
    assert (dw - o["disparity"][..., 0]).abs().mean().item() > 10 * EPE_TOL


@pytest.mark.parametrize("bname,size", SIZES[:2])
@pytest.mark.parametrize("scale,block", [(2, 4), (3, 1)])
def test_mad_step_reprojection_scale(bname, size, scale, block):
    """--reprojectionScale s (Stereo_Online_Adaptation.py:22-23,91-107): the MAD block's loss on the frames and the
    prediction resized to (H//s, W//s); full-resolution loss / metrics unchanged."""
    backend = _backend(bname)
    if bname == "emul" and block == 1:
        pytest.skip("CPU emulator: one block (time)")
    eng, wn, wt, acc, (l, r, gt) = _setup(backend, *size)
    eng.set_reprojection_scale(scale)
    blocks = json.load(open(os.path.join(PKG, "block_config", "MadNet_full.json")))
    lv = OM.layer_variables()
    bv = sum([lv[n] for n in blocks[block]], [])
    lr = 1e-2
    eng.build_plan("MAD", lr=lr, block_vars=bv, block_level=E.LEVELS[block]).run(backend.lib, 0)
    o = OM.step(wt, acc, l, r, gt, mode="MAD", block_vars=bv, block_index=block, lr=lr, reprojection_scale=scale)
    _check(eng, wn, wt, o, backend)


@pytest.mark.gpu
def test_two_host_threads_one_device(hip):
    """Re-entrancy (SURVEY 8(b) "Threading / streams"): two host threads, each with its own Adapter on the SAME GPU, step
    concurrently (hipGraph replay with side lanes, per-thread lane streams / events in the library); both end exactly where
    the same adapters end when stepped one after the other."""
    import threading
    import Nets
    from madnet_hip.adapter import Adapter
    H, W = 128, 256
    wn = S.calibrated_weights(OM.variable_shapes(), 1)
    pairs = [S.make_pair(H, W, stream_id=i) for i in range(2)]

    def make(i):
        l, r, gt = pairs[i]
        z = torch.zeros(1, H, W, 3, device="cuda")
        net = Nets.get_stereo_net("MADNet", {"left_img": z, "right_img": z, "split_layers": [None], "sequence": True, "train_portion": "BEGIN",
                                             "bulkhead": False, "weights": wn, "warping": True, "context_net": True, "radius_d": 2, "stride": 1})
        return Adapter(net, mode="FULL", lr=1e-3), tuple(torch.from_numpy(a).cuda() for a in (l, r, gt[..., 0]))

    def run(ad, data, n, out, key):
        torch.cuda.set_device(0)
        for _ in range(n):
            o = ad.step(*data)
        torch.cuda.synchronize()
        out[key] = (ad.eng.params.w.clone(), o["loss"])

    serial, serial2, conc = {}, {}, {}
    for i in range(2):
        ad, data = make(i)
        run(ad, data, 6, serial, i)
        ad, data = make(i)
        run(ad, data, 6, serial2, i)               # the run-to-run noise of the fp32 atomics (warp / bias gradients) after 6 steps
    ads = [make(i) for i in range(2)]
    for ad, _ in ads:
        ad._plan("FULL")                           # capture both graphs before the threads start
    th = [threading.Thread(target=run, args=(ads[i][0], ads[i][1], 6, conc, i)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    for i in range(2):
        noise_w = (serial[i][0] - serial2[i][0]).abs().max().item()
        noise_l = abs(serial[i][1] - serial2[i][1])
        dw = (serial[i][0] - conc[i][0]).abs().max().item()
        dl = abs(serial[i][1] - conc[i][1])
        print("thread %d: |dw| concurrent-vs-serial %.3g (serial-vs-serial %.3g), |dloss| %.3g (%.3g)" % (i, dw, noise_w, dl, noise_l))
        # same kernels, same order: only the summation order of the fp32 atomics differs, exactly as between two serial runs.  That
        # difference is amplified chaotically over the 6 steps: measured serial-vs-serial samples on one box range from 1.5e-11 to 6e-5
        # (weights) / 0 to 1e-5 (loss), so ONE sample is no yardstick for a 5x bound -- a broken lane / event ring shows up as 1e-2 and more
        assert dl <= max(5 * noise_l, 2e-4 * max(1.0, abs(serial[i][1]))), (i, dl, noise_l)
        assert dw <= max(5 * noise_w, 5e-4), (i, dw, noise_w)


@pytest.mark.gpu
def test_two_steps_and_graph_replay(hip):
    """Second step starts from the updated weights + momentum; the captured hipGraph replays the
    same op array (graph replay == eager plan)."""
    backend = hip
    H, W = 128, 256
    eng, wn, wt, acc, (l, r, gt) = _setup(backend, H, W)
    lr = 1e-3
    plan = eng.build_plan("FULL", lr=lr)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        plan.capture(backend.lib, stream.cuda_stream)
        for _ in range(2):
            plan.launch(backend.lib, stream.cuda_stream)
    stream.synchronize()
    for _ in range(2):
        o = OM.step(wt, acc, l, r, gt, mode="FULL", lr=lr)
    pred = eng.pred.cpu()
    assert (pred - o["disparity"][..., 0]).abs().mean().item() <= EPE_TOL
    for n in wt:
        assert (eng.params.tensor(n).cpu() - wt[n]).abs().max().item() <= 2e-5 * max(1.0, wt[n].abs().max().item()), n


@pytest.mark.gpu
def test_schedules_agree_and_batched_streams(hip):
    """The same FULL step under every scheduling variant (filter gradients: atomics / workspace+reduction, serial / on
    side lanes, eager / hipGraph replay) ends in the same weights up to fp32 summation order -- a race between a side
    lane and the main lane would show up here; and B=2 streams through one engine (shared model) equal the oracle's
    batch-2 step."""
    H, W = 128, 256
    shapes = OM.variable_shapes()
    wn = S.calibrated_weights(shapes, 1)
    l, r, gt = S.make_pair(H, W)
    results = {}
    for name, partial, lanes, graph in (("atomics", False, 0, False), ("partial", True, 0, False), ("lanes", True, 2, False),
                                        ("lanes+graph", True, 2, True)):
        eng = E.MadNetEngine(hip.lib, H, W, B=1, device=hip.device, weights=wn)
        eng.partial_wgrad, eng.wgrad_lanes = partial, lanes
        eng.set_inputs(l, r, gt[..., 0])
        plan = eng.build_plan("FULL", lr=1e-4)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            if graph:
                plan.capture(hip.lib, st.cuda_stream)
            for _ in range(3):                      # three steps: momentum + weights feed back
                plan.launch(hip.lib, st.cuda_stream)
            st.synchronize()
        results[name] = (eng.params.w.clone(), eng.params.m.clone(), float(eng.res_loss[0].item()))
    w0, m0, l0 = results["atomics"]
    for name, (w, m, ls) in results.items():
        assert abs(ls - l0) <= 1e-5 * max(1.0, abs(l0)), name
        assert (w - w0).abs().max().item() <= 2e-6 * max(1.0, w0.abs().max().item()), name
        assert (m - m0).norm().item() <= 2e-3 * max(m0.norm().item(), 1e-12), name
    # batched streams: B = 2 with one model == oracle on the stacked batch
    l2, r2, g2 = S.make_pair(H, W, stream_id=5)
    L, R, G = np.concatenate([l, l2]), np.concatenate([r, r2]), np.concatenate([gt, g2])
    eng = E.MadNetEngine(hip.lib, H, W, B=2, device=hip.device, weights=wn)
    eng.set_inputs(L, R, G[..., 0])
    eng.build_plan("FULL", lr=1e-4).run(hip.lib, 0)
    torch.cuda.synchronize()
    wt = {k: torch.from_numpy(v.copy()) for k, v in wn.items()}
    acc = {k: torch.zeros_like(v) for k, v in wt.items()}
    o = OM.step(wt, acc, torch.from_numpy(L), torch.from_numpy(R), torch.from_numpy(G), mode="FULL", lr=1e-4)
    assert (eng.pred.cpu() - o["disparity"][..., 0]).abs().mean().item() <= EPE_TOL
    assert abs(eng.res_loss[0].item() - o["loss"]) <= 2e-5 * max(1.0, abs(o["loss"]))
    for n in wt:
        assert (eng.params.tensor(n).cpu() - wt[n]).abs().max().item() <= 1e-5 * max(1.0, wt[n].abs().max().item()), n


@pytest.mark.gpu
def test_bf16_mode_end_to_end(hip):
    """Throughput mode: same step with bf16 MFMA inputs.  Bounded deviation from the fp32 engine (documented in every
    bench line as epe_vs_oracle), identical control outputs (loss / metrics finite, weights move the same way)."""
    H, W = 128, 256
    wn = S.calibrated_weights(OM.variable_shapes(), 1)
    l, r, gt = S.make_pair(H, W)
    out = {}
    for prec in ("fp32", "bf16"):
        eng = E.MadNetEngine(hip.lib, H, W, B=1, device=hip.device, weights=wn, precision=prec)
        eng.set_inputs(l, r, gt[..., 0])
        eng.build_plan("FULL", lr=1e-4).run(hip.lib, 0)
        torch.cuda.synchronize()
        out[prec] = (eng.pred.clone(), float(eng.res_loss[0].item()), eng.params.w.clone(), eng.params.g.clone())
    p32, l32, w32, g32 = out["fp32"]; p16, l16, w16, g16 = out["bf16"]
    scale = p32.abs().mean().item()
    assert torch.isfinite(p16).all() and (p16 - p32).abs().mean().item() <= 0.15           # measured 0.076 px at this shape (smoke)
    assert abs(l16 - l32) <= 0.02 * max(abs(l32), 1e-3)
    cos = torch.nn.functional.cosine_similarity(g16.flatten(), g32.flatten(), dim=0).item()
    assert cos >= 0.98, cos                       # the bf16 gradient points the same way
    assert not torch.equal(w16, w32)              # and it really is a different arithmetic


def _mixed_vs_oracle(lib, device, H, W, force_patch, block=None):
    """One FULL step (block = None) or one MAD step on block `block` of MadNet_full.json in the 'mixed' mode (forward: split-bf16 on the layers
    with an x3 kernel, exact fp32 elsewhere; gradients: bf16 MFMA) judged against the fp32 CPU oracle."""
    wn = S.calibrated_weights(OM.variable_shapes(), 1)
    l, r, gt = S.make_pair(H, W)
    lib.tune_conv_patch(force_patch)
    lib.tune_conv_planes(0)
    try:
        eng = E.MadNetEngine(lib, H, W, B=1, device=device, weights=wn, precision="mixed")
        eng.set_inputs(l, r, gt[..., 0])
        lr = 1e-4
        bv = None
        if block is None:
            eng.build_plan("FULL", lr=lr).run(lib, 0)
        else:
            blocks = json.load(open(os.path.join(PKG, "block_config", "MadNet_full.json")))
            lv = OM.layer_variables()
            bv = sum([lv[n] for n in blocks[block]], [])
            eng.build_plan("MAD", lr=lr, block_vars=bv, block_level=E.LEVELS[block]).run(lib, 0)
        if device != "cpu":
            torch.cuda.synchronize()
    finally:
        launches = lib.tune_conv_patch(-1) + lib.tune_conv_planes(0)       # patch-staged / fragment-bank / planes kernels that really ran
    wt = {k: torch.from_numpy(v.copy()) for k, v in wn.items()}
    acc = {k: torch.zeros_like(v) for k, v in wt.items()}
    if block is None:
        o = OM.step(wt, acc, torch.from_numpy(l), torch.from_numpy(r), torch.from_numpy(gt), mode="FULL", lr=lr)
    else:
        o = OM.step(wt, acc, torch.from_numpy(l), torch.from_numpy(r), torch.from_numpy(gt), mode="MAD", block_vars=bv, block_index=block, lr=lr)
    epe = (eng.pred.cpu() - o["disparity"][..., 0]).abs().mean().item()
    gh = torch.cat([eng.params.tensor(n, "g").cpu().flatten() for n in o["grads"]])
    go = torch.cat([g.flatten() for g in o["grads"].values()])
    cos = torch.nn.functional.cosine_similarity(gh, go, dim=0).item()
    grel = (gh - go).norm().item() / go.norm().item()
    dw = max((eng.params.tensor(n).cpu() - wt[n]).abs().max().item() for n in o["grads"])
    step = max((torch.from_numpy(wn[n]) - wt[n]).abs().max().item() for n in o["grads"])          # size of the oracle's own update
    loss_err = abs(eng.res_loss[0].item() - o["loss"])
    untouched = all(torch.equal(eng.params.tensor(n).cpu(), torch.from_numpy(wn[n])) for n in wt if n not in o["grads"])
    return dict(epe=epe, cos=cos, grel=grel, dw=dw, step=step, loss_err=loss_err, loss=o["loss"], launches=launches,
                mean_disp=o["disparity"].abs().mean().item(), untouched=untouched)


def test_mixed_mode_step_emulated():
    """'mixed' on the emulator (60x100, the x3 patch kernel forced onto every eligible forward layer): the forward pass sits
    inside the north-star tolerance with a wide margin, the bf16 gradients point the oracle's way."""
    from conftest import _emul_backend
    backend = _emul_backend()
    m = _mixed_vs_oracle(backend.lib, "cpu", 60, 100, 128)
    print("mixed (emulated 60x100): %s" % m)
    assert m["launches"] >= 20, m                  # x3 forward instances + bf16 input-gradient instances really ran
    assert m["epe"] <= 0.2 * EPE_TOL, m
    assert m["loss_err"] <= 1e-4 * max(1.0, abs(m["loss"])), m
    assert m["cos"] >= 0.98, m
    assert m["dw"] <= 0.25 * m["step"], m          # the post-step weights deviate by a fraction of the step itself


@pytest.mark.gpu
def test_mixed_mode_headline_config_within_tolerance(hip):
    """The bench default (BASELINE config 2, 1242x375): disparity of the 'mixed' step within 1e-3 px of the fp32 oracle --
    the tolerance the north star states -- with the kernels the heuristic dispatches on its own (x3 patch kernel at 1/4
    resolution, exact fp32 elsewhere in the forward pass, bf16 gradients)."""
    m = _mixed_vs_oracle(hip.lib, hip.device, 375, 1242, -1)
    print("mixed (MI355X 375x1242): %s" % m)
    assert m["launches"] >= 10, m
    assert m["epe"] <= EPE_TOL, m
    assert m["loss_err"] <= 1e-4 * max(1.0, abs(m["loss"])), m
    # the bf16 gradients: 4 x what scripts/measure_mixed_parity.py measured on the MI355X (round 3: relative L2 of the whole gradient 3.5e-3, largest
    # post-step weight deviation 3.4e-3 of the step itself) -- a broken bf16 filter-gradient instance moves these by orders of magnitude
    assert m["cos"] >= 0.999 and m["grel"] <= 1.5e-2 and m["dw"] <= 1.5e-2 * m["step"], m


@pytest.mark.gpu
@pytest.mark.parametrize("block,grel_max,dw_max", [(0, 9e-2, 7e-2), (4, 1.5e-2, 1.5e-2)])
def test_mad_step_mixed_mode_headline_config(hip, block, grel_max, dw_max):
    """The MAD bench line runs 'mixed': one MAD step on the coarsest (0: estimator 6 + conv11/12) and the finest block (4: estimator 2 + conv1-4 + the
    context network) at 1242x375 against the fp32 oracle -- disparity inside the tolerance, the block's gradients within 4 x the measured deviation
    (block 0: relative L2 2.2e-2, update 1.8e-2 of the step; block 4: 3.6e-3 / 3.6e-3), every variable outside the block BIT-identical."""
    m = _mixed_vs_oracle(hip.lib, hip.device, 375, 1242, -1, block=block)
    print("mixed MAD block %d (MI355X 375x1242): %s" % (block, m))
    assert m["epe"] <= EPE_TOL and m["untouched"], m
    assert m["loss_err"] <= 1e-4 * max(1.0, abs(m["loss"])), m
    assert m["cos"] >= 0.999 and m["grel"] <= grel_max and m["dw"] <= dw_max * m["step"], m


@pytest.mark.gpu
def test_bf16_mode_headline_config_bounded(hip):
    """Plain bf16 at the headline shape, where the patch-staged kernel is dispatched by the heuristic: NOT within the 1e-3
    tolerance (that is what 'mixed' is for) -- bounded at 2x the measured 0.083 px (mean |d| 9.5 px)."""
    wn = S.calibrated_weights(OM.variable_shapes(), 1)
    l, r, gt = S.make_pair(375, 1242)
    eng = E.MadNetEngine(hip.lib, 375, 1242, B=1, device=hip.device, weights=wn, precision="bf16")
    eng.set_inputs(l, r, gt[..., 0])
    hip.lib.tune_conv_patch(-1)
    eng.build_plan("NONE").run(hip.lib, 0)
    torch.cuda.synchronize()
    assert hip.lib.tune_conv_patch(-1) >= 5
    with torch.no_grad():
        d = OM.forward({k: torch.from_numpy(v) for k, v in wn.items()}, torch.from_numpy(l), torch.from_numpy(r))[-1][..., 0]
    epe = (eng.pred.cpu() - d).abs().mean().item()
    print("bf16 (MI355X 375x1242) EPE vs oracle %.4f" % epe)
    assert 1e-3 < epe <= 0.17


def test_bf16_step_patch_kernel_vs_gather_kernel_emulated():
    """The patch-staged conv kernel inside the whole engine (channel-slice views, fused masks, accumulating gradients): one bf16
    FULL step with the kernel forced onto every eligible layer (mh_tune_conv_patch(128 + 256) bypasses the size heuristic, so the
    60x100 emulator case runs it at every level) against the same step on the gather kernel.  Both round their operands to
    bf16 identically; only the fp32 summation order differs."""
    from conftest import _emul_backend
    backend = _emul_backend()
    H, W = 60, 100
    wn = S.calibrated_weights(OM.variable_shapes(), 1)
    l, r, gt = S.make_pair(H, W)
    out = {}
    for mode in (0, 128 + 256, "fp32"):
        backend.lib.tune_conv_patch(0 if mode == "fp32" else mode)
        try:
            eng = E.MadNetEngine(backend.lib, H, W, B=1, device="cpu", weights=wn, precision="fp32" if mode == "fp32" else "bf16")
            eng.use_bank = False          # (the fragment-bank kernels would take the small layers: they have their own test below)
            eng.set_inputs(l, r, gt[..., 0])
            eng.build_plan("FULL", lr=1e-4).run(backend.lib, 0)
        finally:
            launches = backend.lib.tune_conv_patch(-1)
        out[mode] = (eng.pred.clone(), float(eng.res_loss[0].item()), eng.params.g.clone(), launches)
    p0, l0, g0, n0 = out[0]; p1, l1, g1, n1 = out[128 + 256]; p32, l32, g32, _ = out["fp32"]
    assert n0 == 0 and n1 >= 40, (n0, n1)                 # estimators 2-6 + context network, forward and input gradients
    # A different fp32 summation order flips the bf16 rounding of an activation now and then (1 bf16 ulp = 0.4 %) and the next
    # layers amplify it, so two bf16 runs agree only at the level of the bf16 noise itself.  The yardstick is therefore the fp32
    # engine: the patch-kernel step must sit as close to it as the gather-kernel step does, and the two bf16 steps must be
    # closer to each other than their two deviations combined.
    dev0, dev1, mutual = (p0 - p32).abs().mean().item(), (p1 - p32).abs().mean().item(), (p1 - p0).abs().mean().item()
    gd0 = (g0 - g32).norm().item() / g32.norm().item(); gd1 = (g1 - g32).norm().item() / g32.norm().item()
    gm = (g1 - g0).norm().item() / g32.norm().item()
    print("bf16 vs fp32 (emulated 60x100): disparity dev gather %.3g patch %.3g mutual %.3g; gradient dev gather %.3g patch %.3g mutual %.3g; "
          "loss %.6f / %.6f / fp32 %.6f; patch launches %d" % (dev0, dev1, mutual, gd0, gd1, gm, l0, l1, l32, n1))
    assert dev1 <= 1.5 * dev0 + 1e-3 and mutual <= dev0 + dev1
    assert gd1 <= 1.5 * gd0 + 1e-3 and gm <= gd0 + gd1
    assert abs(l1 - l32) <= 1.5 * abs(l0 - l32) + 1e-3 * abs(l32)


@pytest.mark.parametrize("precision", ["bf16", "mixed"])
def test_step_fragment_bank_kernels_vs_tiled_kernels_emulated(precision):
    """The fragment-bank kernels inside the whole engine (mh_pack_weights at the start of the plan, forward and input-gradient banks, channel-
    slice views, fused masks, accumulating gradients): one FULL step with the banks on (at 60x100 every 3x3 stride-1 layer is a "small"
    layer -> conv_bank_small_kernel; the split-bf16 ones of the 'mixed' mode too) against the same step on the tiled kernels.  Same
    operand rounding, another summation order -> judged like the patch kernel above: against the fp32 engine."""
    from conftest import _emul_backend
    backend = _emul_backend()
    H, W = 60, 100
    wn = S.calibrated_weights(OM.variable_shapes(), 1)
    l, r, gt = S.make_pair(H, W)
    out = {}
    for mode in ("tiled", "bank", "fp32"):
        backend.lib.tune_conv_bank(-1)
        eng = E.MadNetEngine(backend.lib, H, W, B=1, device="cpu", weights=wn, precision="fp32" if mode == "fp32" else precision)
        eng.use_bank = mode == "bank"
        eng.set_inputs(l, r, gt[..., 0])
        eng.build_plan("FULL", lr=1e-4).run(backend.lib, 0)
        out[mode] = (eng.pred.clone(), float(eng.res_loss[0].item()), eng.params.g.clone(), backend.lib.tune_conv_bank(-1))
    p0, l0, g0, n0 = out["tiled"]; p1, l1, g1, n1 = out["bank"]; p32, l32, g32, _ = out["fp32"]
    assert n0 == 0 and n1 >= 30, (n0, n1)          # estimators + context + stride-1 pyramid layers, forward and input gradients (the 1x2 .. 4x7 levels stay on the tiled kernel: tile cover)
    dev0, dev1, mutual = (p0 - p32).abs().mean().item(), (p1 - p32).abs().mean().item(), (p1 - p0).abs().mean().item()
    gd0 = (g0 - g32).norm().item() / g32.norm().item(); gd1 = (g1 - g32).norm().item() / g32.norm().item()
    gm = (g1 - g0).norm().item() / g32.norm().item()
    print("%s vs fp32 (emulated 60x100): disparity dev tiled %.3g bank %.3g mutual %.3g; gradient dev tiled %.3g bank %.3g mutual %.3g; "
          "loss %.6f / %.6f / fp32 %.6f; bank launches %d" % (precision, dev0, dev1, mutual, gd0, gd1, gm, l0, l1, l32, n1))
    assert dev1 <= 1.5 * dev0 + 1e-3 and mutual <= dev0 + dev1 + 1e-4
    assert gd1 <= 1.5 * gd0 + 1e-3 and gm <= gd0 + gd1 + 1e-4
    assert abs(l1 - l32) <= 1.5 * abs(l0 - l32) + 1e-3 * abs(l32)


def _proxy_from(gt, seed=3):
    """Proxy labels = ground truth + noise, with holes (<= 0) and out-of-range values (>= 192) as the validity test expects."""
    g = torch.Generator().manual_seed(seed)
    px = gt[..., 0] + torch.randn(gt.shape[:-1], generator=g) * 1.5
    px[torch.rand(px.shape, generator=g) < 0.3] = 0.0
    px[0, :2, :5] = 200.0
    return px


@pytest.mark.parametrize("bname,size", [SIZES[0], pytest.param("hip", (128, 256), marks=pytest.mark.gpu, id="hip-128x256")])
@pytest.mark.parametrize("mode,block", [("FULL", None), ("MAD", 4), ("MAD", 1)])
def test_continual_proxy_step(bname, size, mode, block):
    """The continual-adaptation variant (Stereo_Continual_Adaptation.py): proxy-label mean_l1 loss, weight 0.01 on the
    full loss / 0.1 on a MAD block's loss, same backward + momentum machinery."""
    backend = _backend(bname)
    if bname == "emul" and block == 1:
        pytest.skip("CPU emulator: one MAD block (time)")
    eng, wn, wt, acc, (l, r, gt) = _setup(backend, *size)
    px = _proxy_from(gt)
    eng.loss_kind = "proxy"
    eng.set_inputs(l, r, gt[..., 0], proxy=px)
    lr = 1e-2
    if mode == "FULL":
        plan = eng.build_plan("FULL", lr=lr)
        o = OM.step(wt, acc, l, r, gt, mode="FULL", lr=lr, loss="proxy", proxy=px[..., None])
    else:
        blocks = json.load(open(os.path.join(PKG, "block_config", "MadNet_full.json")))
        lv = OM.layer_variables()
        bv = sum([lv[n] for n in blocks[block]], [])
        plan = eng.build_plan("MAD", lr=lr, block_vars=bv, block_level=E.LEVELS[block])
        o = OM.step(wt, acc, l, r, gt, mode="MAD", block_vars=bv, block_index=block, lr=lr, loss="proxy", proxy=px[..., None])
    plan.run(backend.lib, 0)
    _check(eng, wn, wt, o, backend)


@pytest.mark.parametrize("bname,size", [SIZES[0], pytest.param("hip", (128, 256), marks=pytest.mark.gpu, id="hip-128x256")])
def test_offline_training_step(bname, size):
    """8(f)-4, Train.py:94-102: one offline training step = forward without bulkhead, multi-scale supervised mean_l1 on all
    six predictions (distinct weights per scale), gradients of every variable, Adam -- two consecutive steps (the second one
    exercises the advanced beta powers and non-zero moments) against the oracle's autograd + fp32 Adam restatement."""
    backend = _backend(bname)
    H, W = size
    eng, wn, wt, _, (l, r, gt) = _setup(backend, H, W, seed=5)
    lw = [1.0, 0.8, 0.6, 0.4, 0.2, 0.1]
    plan = eng.build_plan("TRAIN", lr=1e-3, loss_weights=lw, max_disp=192.0)
    am = {k: torch.zeros_like(v) for k, v in wt.items()}
    av = {k: torch.zeros_like(v) for k, v in wt.items()}
    st = [0.9, 0.999]
    for step in range(2):
        plan.run(backend.lib, 0)
        backend.sync()
        o = OM.train_step(wt, am, av, st, l, r, gt, lr=1e-3, loss_weights=lw, max_disp=192.0)
        assert (eng.pred.cpu() - o["disparity"][..., 0]).abs().mean().item() <= EPE_TOL
        got = eng.res_loss_ms[:, 0].cpu().tolist()
        for i, (a, b) in enumerate(zip(got, o["losses"])):
            assert abs(a - b) <= 5e-5 * max(1.0, abs(b)), (step, i, a, b)
        if step == 0:                                   # gradients: same criteria as the adaptation steps
            gmax = max(g.abs().max().item() for g in o["grads"].values())
            for n, g in o["grads"].items():
                ge = eng.params.tensor(n, "g").cpu()
                rel = (ge - g).norm().item() / max(g.norm().item(), 1e-30)
                # (|x| losses: a prediction within rounding of its target flips the sign of that pixel's gradient, so the
                #  agreement of the filter gradients degrades with fewer pixels: 2e-3 holds from 60x100 up)
                assert rel <= 2e-3 or g.abs().max().item() <= 1e-6 * gmax, (n, rel)
        # Adam normalises every element's step to ~lr whatever the gradient's size (first step: lr * g / (|g| + 1e-8) = +-lr): an
        # element whose gradient is at the rounding level of its tensor takes a full step in a direction the fp32 summation ORDER
        # decides (4-lane split of the resize gradient, pixel splits of the filter gradients ...), and at step 1 the network itself
        # then differs.  So (a) the worst-element bound is taken over the elements whose oracle gradient is above that floor
        # (|g| > 1e-4 max|g| of the tensor), all elements count for the mean bound; (b) after every step the oracle continues from
        # the ENGINE's weights and Adam moments, so that step 1 tests the step-1 arithmetic (advanced beta powers, non-zero
        # moments) instead of the amplified rounding noise of step 0.
        # (c) the |x| losses again: a prediction within rounding of its target flips the sign of that pixel's gradient, and at the coarse
        # levels of a 60x100 image (2x4 .. 8x13 pixels) one pixel is a visible share of a layer's gradient -- a handful of elements of ONE
        # tensor then step the other way (seen when the summation order of the heads' input gradient changed: 7 of 109272 elements of
        # conv9).  So the worst-element bound holds for all but 1e-3 of a tensor's solid elements, and nothing moves further than Adam can
        # push it in one step (2.5 lr).
        worst, mean, frac = 0.0, 0.0, 0.0
        for n in wt:
            d = (eng.params.tensor(n).cpu() - wt[n]).abs()
            g = o["grads"].get(n)
            solid = (g.abs() > 1e-4 * g.abs().max()) if g is not None else torch.ones_like(d, dtype=torch.bool)
            if solid.any():
                worst = max(worst, d[solid].max().item())
                frac = max(frac, (d[solid] > 0.25 * 1e-3).float().mean().item())
            mean = max(mean, d.mean().item())
        print("offline step %d: worst |dw| (solid gradients) %.3g lr, share of a tensor's solid elements beyond 0.25 lr %.2g, worst tensor-mean %.3g lr"
              % (step, worst / 1e-3, frac, mean / 1e-3))
        assert frac <= 1e-3 and worst <= 2.5 * 1e-3 and mean <= 1e-3 * 1e-3, (step, worst, frac, mean)
        for n in wt:
            wt[n] = eng.params.tensor(n).cpu().clone()
            am[n] = eng.params.tensor(n, "m").cpu().clone(); av[n] = eng.params.tensor(n, "v").cpu().clone()
        assert torch.allclose(eng.adam_state.cpu(), torch.tensor(st), rtol=1e-6)



def test_plan_scheduling_switches_do_not_change_results_emulated(monkeypatch):
    """The scheduling knobs of a plan -- undeferred side batches (MH_OP_NODEFER), two filter-gradient lanes, no side loss, per-level fills instead of the
    single one, unfused level backward -- only move launches between lanes / change their order: on the emulator (no atomics races: workgroups run one
    after the other) every variant must reproduce the default plan's forward pass bit for bit and its updated weights to the order of the fp32 atomics."""
    from conftest import _emul_backend
    backend = _emul_backend()
    H, W = 60, 100
    wn = S.calibrated_weights(OM.variable_shapes(), 1)
    l, r, gt = S.make_pair(H, W)

    def run(**kw):
        eng = E.MadNetEngine(backend.lib, H, W, B=1, device="cpu", weights=wn, precision="mixed", schedule=E.Schedule(**{k: v for k, v in kw.items() if k != "lanes"}))
        if "lanes" in kw:
            eng.wgrad_lanes = kw["lanes"]
        eng.set_inputs(l, r, gt[..., 0])
        eng.build_plan("FULL", lr=1e-4).run(backend.lib, 0)
        return eng.pred.clone(), float(eng.res_loss[0].item()), eng.params.w.clone()

    p0, l0, w0 = run()
    for kw, exact in (({"NODEFER_BATCHES": 3}, True), ({"lanes": 2}, True), ({"lanes": 0}, True), ({"SIDE_LOSS": False}, True),
                      ({"ONE_FILL": False}, False), ({"FUSE_BACK": False}, False), ({"HEAD_IN_FRONT": False}, "head"), ({"IMAGE_CONV": False}, "image")):
        p1, l1, w1 = run(**kw)
        if exact == "image":
            # conv1 on the padded copy (pad_reflect + the row kernel, split-bf16) instead of straight from the frames (exact fp32): other arithmetic in ONE layer
            # (2^-16 relative per product) -- the step must agree far inside the oracle tolerance
            assert (p0 - p1).abs().mean().item() <= 2e-4 and abs(l0 - l1) <= 1e-5 * abs(l0) and (w0 - w1).abs().max().item() <= 1e-6, (kw, (p0 - p1).abs().mean().item(), l0, l1)
            continue
        if exact == "head":
            # the disparity heads of levels 6 .. 3 as launches of their own: the same arithmetic in another kernel -- on the emulator (one compiler, no fma
            # contraction differences) the coarse disparities, hence everything downstream, must come out bit for bit
            assert torch.equal(p0, p1) and l0 == l1 and (w0 - w1).abs().max().item() <= 1e-10, (kw, (p0 - p1).abs().max().item(), (w0 - w1).abs().max().item())
            continue
        assert torch.equal(p0, p1) and l0 == l1, kw
        # scheduling only: the same kernels on the same data (the fp32 atomics of the bias / warp gradients may land in another order: ~1e-13);
        # ONE_FILL / FUSE_BACK swap kernels (accumulate onto zero, fused correlation + warp gradient) = another fp32 summation order
        assert (w0 - w1).abs().max().item() <= (1e-10 if exact else 1e-7), (kw, (w0 - w1).abs().max().item())


@pytest.mark.parametrize("bname,size", [SIZES[0], pytest.param("hip", (128, 256), marks=pytest.mark.gpu, id="hip-128x256")])
@pytest.mark.parametrize("mode,block", [("FULL", None), ("MAD", 4)])
def test_live_demo_adaptation_step_adam(bname, size, mode, block):
    """8(f)-4, Demo/demo_model.py:110-164,233-250: the live demo runs the SAME forward / reprojection loss / FULL or per-block backward as the online
    script but applies tf.train.AdamOptimizer(lr) -- one optimizer object for every train op, so per-variable slots and one pair of beta powers that advances
    with every executed train op.  Two consecutive steps against the oracle (autograd + the fp32 ApplyAdam restatement); criteria as in the offline step."""
    backend = _backend(bname)
    H, W = size
    eng, wn, wt, acc, (l, r, gt) = _setup(backend, H, W, seed=3)
    lr = 1e-3
    bv = None
    if mode == "MAD":
        blocks = json.load(open(os.path.join(PKG, "block_config", "MadNet_full.json")))
        lv = OM.layer_variables()
        bv = sum([lv[n] for n in blocks[block]], [])
        plan = eng.build_plan("MAD", lr=lr, block_vars=bv, block_level=E.LEVELS[block], optimizer="adam")
    else:
        plan = eng.build_plan("FULL", lr=lr, optimizer="adam")
    with pytest.raises(ValueError):
        eng.build_plan("FULL", lr=lr, optimizer="rmsprop")
    adam = {"m": {k: torch.zeros_like(v) for k, v in wt.items()}, "v": {k: torch.zeros_like(v) for k, v in wt.items()}, "state": [0.9, 0.999]}
    for step in range(2):
        plan.run(backend.lib, 0)
        backend.sync()
        o = OM.step(wt, acc, l, r, gt, mode=mode, block_vars=bv, block_index=block, lr=lr, adam=adam)
        assert (eng.pred.cpu() - o["disparity"][..., 0]).abs().mean().item() <= EPE_TOL
        assert abs(eng.res_loss[0].item() - o["loss"]) <= 2e-5 * max(1.0, abs(o["loss"]))
        worst, mean, nsolid, nall, dsum = 0.0, 0.0, 0, 0, 0.0
        for n in wt:
            we = eng.params.tensor(n).cpu()
            g = o["grads"].get(n)
            if g is None:                                  # MAD: outside the sampled block nothing moves, slots stay zero
                assert torch.equal(we, torch.from_numpy(wn[n])), n
                assert not eng.params.tensor(n, "m").any() and not eng.params.tensor(n, "v").any(), n
                continue
            d = (we - wt[n]).abs()
            # Adam normalises every element's step to ~lr (see test_offline_training_step); from the second step on the step is LINEAR in the gradient's
            # relative error (lr_t (1-b1) dg / sqrt(v) ~ 0.5 lr dg/|g|), and the reprojection loss' gradients (SSIM + |x| kinks, warp) carry more
            # summation-order noise than the supervised ones.  So the element bound is taken over the elements whose ENGINE gradient agrees with the oracle's
            # to 5 % -- which must be nearly all of them -- and the tensor-mean bound over everything (MI355X, 128x256, step 0: 98.0 % agree, worst tensor mean
            # 5.9e-3 lr = 0.3 % of a tensor's elements taking the +-lr step in the other direction).
            ge = eng.params.tensor(n, "g").cpu()
            solid = (ge - g).abs() <= 0.05 * g.abs()
            nsolid += int(solid.sum()); nall += solid.numel()
            if solid.any():
                worst = max(worst, d[solid].max().item())
            mean = max(mean, d.mean().item()); dsum += d.sum().item()
        print("demo adam step %d: worst |dw| %.3g lr, worst tensor-mean %.3g lr, mean over all trained elements %.3g lr, %.4f of the elements within 5 %% gradient "
              "agreement" % (step, worst / lr, mean / lr, dsum / max(nall, 1) / lr, nsolid / max(nall, 1)))
        # (a single small tensor can have a few per cent of floor-level elements -- 3.4e-2 lr tensor mean on the MI355X for block 4 -- so the mean bound is global)
        assert worst <= 0.1 * lr and dsum <= 1e-2 * lr * nall and mean <= 0.1 * lr and nsolid >= 0.95 * nall, (step, worst, mean, dsum, nsolid, nall)
        for n in o["grads"]:                               # continue from the engine's state (step 1 tests the step-1 arithmetic)
            wt[n] = eng.params.tensor(n).cpu().clone()
            adam["m"][n] = eng.params.tensor(n, "m").cpu().clone(); adam["v"][n] = eng.params.tensor(n, "v").cpu().clone()
        assert torch.allclose(eng.adam_state.cpu(), torch.tensor(adam["state"]), rtol=1e-6)
    assert torch.allclose(eng.adam_state.cpu(), torch.tensor([0.9 ** 3, 0.999 ** 3]), rtol=1e-5)


@pytest.mark.parametrize("mode", ["FULL", "MAD"])
def test_step_with_streamed_filter_gradients_emulated(mode):
    """The streaming filter-gradient kernel (mh_shadow_cast + mh_wgrad_stream: every stride-1 3x3 layer of a backward batch in one launch, estimators
    at all five levels, the dilated context layers, the N = 1 heads, the batch-2B pyramid layers) inside a whole bf16 step against the same step on
    the tiled kernels: same bf16-rounded operands, so the gradients agree to the fp32 summation order -- except the bias gradients, which the streamed
    path sums from the bf16 shadow of dz (relative 2^-9 per element)."""
    from conftest import _emul_backend
    backend = _emul_backend()
    H, W = 60, 100
    wn = S.calibrated_weights(OM.variable_shapes(), 1)
    l, r, gt = S.make_pair(H, W)
    res = {}
    for stream in (False, True):
        eng = E.MadNetEngine(backend.lib, H, W, B=1, device=backend.device, weights=wn, precision="bf16")
        eng.use_stream = stream
        eng.set_inputs(l, r, gt[..., 0])
        if mode == "FULL":
            plan = eng.build_plan("FULL", lr=1e-2)
        else:
            blocks = json.load(open(os.path.join(PKG, "block_config", "MadNet_full.json")))
            lv = OM.layer_variables()
            plan = eng.build_plan("MAD", lr=1e-2, block_level=E.LEVELS[4], block_vars=sum([lv[n] for n in blocks[4]], []))
        kinds = [o.kind for o in plan.arr]
        plan.run(backend.lib, 0)
        backend.sync()
        eng.last_plan_arr = plan.arr
        res[stream] = (eng.params.w.clone(), eng.params.g.clone(), eng.pred.clone(), kinds, eng)
    from madnet_hip import _ffi
    assert _ffi.OP_WGRAD_STREAM not in res[False][3] and res[True][3].count(_ffi.OP_WGRAD_STREAM) >= (5 if mode == "FULL" else 1)
    (w0, g0, p0, _, e0), (w1, g1, p1, _, e1) = res[False], res[True]
    assert torch.equal(p0, p1)                                              # the forward pass is untouched
    P = e0.params
    for name, _shape in P.manifest:
        a, b = P.tensor(name, "g"), e1.params.tensor(name, "g")
        sc = max(a.abs().max().item(), 1e-6)
        # (the single-output-channel heads ran on wgrad_n1_kernel with UNROUNDED fp32 operands before: streamed, they see bf16 operands like every other layer)
        tol = 2e-5 if (name.endswith("/weights") and _shape[-1] > 1) else 6e-3
        assert (a - b).abs().max().item() <= tol * sc, (name, (a - b).abs().max().item() / sc)
    # shadows written by the producing kernels' epilogues (mh_conv2d_sh, the default) against shadows cast in a separate pass: the same bf16 values,
    # so every gradient is bit-identical
    engc = E.MadNetEngine(backend.lib, H, W, B=1, device=backend.device, weights=wn, precision="bf16")
    engc.fuse_shadows = False
    engc.set_inputs(l, r, gt[..., 0])
    if mode == "FULL":
        planc = engc.build_plan("FULL", lr=1e-2)
    else:
        planc = engc.build_plan("MAD", lr=1e-2, block_level=E.LEVELS[4], block_vars=sum([lv[n] for n in blocks[4]], []))
    planc.run(backend.lib, 0)
    backend.sync()
    for name, _shape in P.manifest:        # (the bias gradients meet in fp32 atomics whose order is free)
        a, b = engc.params.tensor(name, "g"), e1.params.tensor(name, "g")
        assert torch.equal(a, b) if name.endswith("/weights") else (a - b).abs().max().item() <= 1e-6 * max(a.abs().max().item(), 1e-6), name
    assert sum(o.i[0] for o in planc.arr if o.kind == _ffi.OP_SHADOW_CAST) > sum(o.i[0] for o in res[True][4].last_plan_arr if o.kind == _ffi.OP_SHADOW_CAST)


def test_step_with_fused_head_backward_emulated():
    """mh_head_bwd / mh_conv2d_head inside the engine: the FULL step with the heads' output gradient + input gradient in one launch (5 launches: level 2 from dfinal and
    the context input's gradient, levels 3 .. 6 through the resize gradient) against the step on the separate launches (fp32 engine: same arithmetic,
    one summation differs in its order)."""
    from conftest import _emul_backend
    backend = _emul_backend()
    H, W = 60, 100
    wn = S.calibrated_weights(OM.variable_shapes(), 1)
    l, r, gt = S.make_pair(H, W)
    out = {}
    try:
        for fused in (True, False):
            eng = E.MadNetEngine(backend.lib, H, W, B=1, device="cpu", weights=wn, precision="fp32", schedule=E.Schedule(FUSE_HEAD=fused))
            eng.set_inputs(l, r, gt[..., 0])
            plan = eng.build_plan("FULL", lr=1e-4, update=False)
            n_head = sum(1 for k in range(plan.n) if plan.arr[k].kind == _ffi_mod().OP_HEAD_BWD)
            plan.run(backend.lib, 0)
            out[fused] = (eng.params.g.clone(), n_head, plan.n)
    finally:
        pass
    (g1, n1, ops1), (g0, n0, ops0) = out[True], out[False]
    # backward: 2 copies + 4 resize gradients + 5 head input gradients -> 5 launches; forward: the level-2 head stores its result in the context
    # input and in `final` itself (mh_conv2d_head): 2 copies less
    assert n1 == 5 and n0 == 0 and ops1 == ops0 - 8
    assert ((g1 - g0).norm() / g0.norm()).item() <= 1e-5


def _ffi_mod():
    from madnet_hip import _ffi
    return _ffi


@pytest.mark.parametrize("bname,size", [SIZES[0], pytest.param("hip", (128, 256), marks=pytest.mark.gpu, id="hip-128x256")])
def test_private_models_as_branches_of_one_graph(bname, size):
    """mh_plans_run / plan.MultiPlan (several private-model streams on one GPU): the FULL steps of S = 3 engines -- their own weights,
    their own frames (S = 2 on the emulator) -- run as parallel branches (captured into ONE hipGraph on the GPU), two steps; every engine ends where it ends when its plan
    runs alone.  Same kernels on the same data: equal up to the landing order of the fp32 atomics (bias / warp gradients)."""
    from madnet_hip.plan import MultiPlan
    backend = _backend(bname)
    H, W = size
    S_ = 3 if bname == "hip" else 2          # (the emulator runs a FULL step in ~4 s)
    shapes = OM.variable_shapes()

    def make(i):
        wn = S.calibrated_weights(shapes, 1 + i)
        l, r, gt = S.make_pair(H, W, stream_id=7 * i)
        eng = E.MadNetEngine(backend.lib, H, W, B=1, device=backend.device, weights=wn, precision="mixed")
        eng.set_inputs(l, r, gt[..., 0])
        return eng

    alone = []
    for i in range(S_):
        eng = make(i)
        plan = eng.build_plan("FULL", lr=1e-3)
        for _ in range(2):
            plan.run(backend.lib, 0)
        backend.sync()
        alone.append((eng.params.w.clone(), eng.pred.clone(), float(eng.res_loss[0].item())))
    engines = [make(i) for i in range(S_)]
    for e in engines:
        e.wgrad_lanes = 0            # serial chains: a fork inside a forked branch crashes hipStreamEndCapture on ROCm 7.2 (profiles/r03_experiments.txt #15)
    mp = MultiPlan([e.build_plan("FULL", lr=1e-3) for e in engines])
    if bname == "hip":
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            mp.capture(backend.lib, st.cuda_stream)
            for _ in range(2):
                mp.launch(backend.lib, st.cuda_stream)
        st.synchronize()
    else:
        for _ in range(2):
            mp.run(backend.lib, 0)
    backend.sync()
    for i, eng in enumerate(engines):
        w0, p0, l0 = alone[i]
        assert (eng.params.w - w0).abs().max().item() <= 1e-6 * max(1.0, w0.abs().max().item()), i
        assert (eng.pred - p0).abs().max().item() <= 1e-4 and abs(float(eng.res_loss[0].item()) - l0) <= 1e-6, i
    # the streams really are different problems
    assert (alone[0][1] - alone[1][1]).abs().mean().item() > 1e-2


def test_multi_adapter_private_streams_emulated():
    """adapter.MultiAdapter on the CPU emulator (the GPU version with a captured graph: tests/test_api_gpu.py): a FULL stream and a MAD stream with
    private models advance together through mh_plans_run; losses, EPEs, sampled blocks and final weights equal those of the two Adapters run alone."""
    from conftest import _emul_backend
    from madnet_hip.adapter import Adapter, MultiAdapter
    import Nets
    be = _emul_backend()
    H, W, lr, steps = 48, 64, 1e-3, 2
    blocks_cfg = json.load(open(os.path.join(PKG, "block_config", "MadNet_full.json")))
    frames = [[S.make_pair(H, W, frame=t, stream_id=sid) for t in range(steps)] for sid in range(2)]

    def make(sid):
        wn = S.calibrated_weights(OM.variable_shapes(), 1 + sid)
        left = torch.zeros(1, H, W, 3); right = torch.zeros_like(left)
        net = Nets.get_stereo_net("MADNet", {"left_img": left, "right_img": right, "split_layers": [None], "sequence": True, "train_portion": "BEGIN",
                                              "bulkhead": sid == 1, "weights": wn, "precision": "mixed", "_lib": be.lib, "_device": "cpu"})
        if sid == 0:
            return net, Adapter(net, mode="FULL", lr=lr, ssim_th=10.0, use_graph=False)
        return net, Adapter(net, mode="MAD", block_config=blocks_cfg, lr=lr, sample_mode="SEQUENTIAL", num_blocks=1, ssim_th=10.0, use_graph=False)

    alone = []
    for sid in range(2):
        net, ad = make(sid)
        alone.append(([ad.step(fr[0], fr[1], fr[2][..., 0]) for fr in frames[sid]], net.engine.params.w.clone()))
    pairs = [make(sid) for sid in range(2)]
    multi = MultiAdapter([ad for _, ad in pairs])
    got = [multi.step([(frames[sid][t][0], frames[sid][t][1], frames[sid][t][2][..., 0]) for sid in range(2)]) for t in range(steps)]
    assert len(multi._graphs) == 2
    for sid in range(2):
        for t in range(steps):
            a, b = alone[sid][0][t], got[t][sid]
            assert a["blocks"] == b["blocks"] and abs(a["loss"] - b["loss"]) <= 1e-7 and abs(a["epe"] - b["epe"]) <= 1e-6
        assert (alone[sid][1] - pairs[sid][0].engine.params.w).abs().max().item() <= 1e-10


def test_adapter_frames_through_the_input_table_emulated():
    """Adapter.step with frames that already are tensors on the engine's device -- float32 and uint8 -- reads them through the step's first node (mh_fetch_inputs, the host
    rewrites the table every step); host arrays take the copies in front of the step; fetch_inputs=False never uses the table.  Same losses, same weights, bit for bit; a
    frame handed over WITHOUT ground truth keeps the previous one (entry left empty)."""
    from conftest import _emul_backend
    from madnet_hip.adapter import Adapter
    from madnet_hip import _ffi
    import Nets
    be = _emul_backend()
    H, W, steps = 48, 64, 3
    frames = [S.make_pair(H, W, frame=t) for t in range(steps)]

    def run(kind, fetch):
        wn = S.calibrated_weights(OM.variable_shapes(), 1)
        left = torch.zeros(1, H, W, 3); right = torch.zeros_like(left)
        net = Nets.get_stereo_net("MADNet", {"left_img": left, "right_img": right, "weights": wn, "precision": "mixed", "_lib": be.lib, "_device": "cpu"})
        ad = Adapter(net, mode="FULL", lr=1e-3, ssim_th=10.0, use_graph=False, fetch_inputs=fetch)
        out = []
        for t, (l, r, g) in enumerate(frames):
            if kind == "numpy":
                f = (l, r, g[..., 0])
            elif kind == "f32":
                f = (torch.from_numpy(l).reshape(1, H, W, 3), torch.from_numpy(r).reshape(1, H, W, 3), torch.from_numpy(np.ascontiguousarray(g[..., 0])))
            else:
                f = (torch.from_numpy(l.astype(np.uint8)), torch.from_numpy(r.astype(np.uint8)), torch.from_numpy(np.ascontiguousarray(g[..., 0])))
            if t == 2:
                f = f[:2]                        # no ground truth this frame: the previous one stays
            out.append(ad.step(*f))
        plan = ad._plan("FULL")[0]
        has_fetch = any(plan.arr[i].kind == _ffi.OP_FETCH_INPUTS for i in range(plan.n))
        return [(o["loss"], o["epe"]) for o in out], net.engine.params.w.clone(), has_fetch

    ref, w_ref, hf = run("numpy", False)
    assert not hf
    for kind in ("f32", "u8"):                   # (host arrays with the table present: every other Adapter test)
        got, w, hf = run(kind, True)
        assert hf and got == ref and torch.equal(w, w_ref), (kind, got, ref)


@pytest.mark.parametrize("bname,size", [SIZES[0], pytest.param("hip", (375, 1242), marks=pytest.mark.gpu, id="hip-375x1242")])
def test_step_with_bf16_only_gradient_maps(bname, size):
    """engine._elide_fp32_gradient_maps: an input gradient whose fp32 result only the next (shadow-staging) input gradient would read stores just the
    bf16 shadow, and the leaky masks test the sign of the activations' shadows (mh_conv2d_sh3).  Same filter gradients as the step that writes every
    fp32 map -- with the fp32 gradient maps poisoned beforehand, so a kernel that still read one would show."""
    backend = _backend(bname)
    H, W = size
    wn = S.calibrated_weights(OM.variable_shapes(), 1)
    l, r, gt = S.make_pair(H, W)
    if bname == "emul":
        backend.lib.tune_conv_patch(128)            # at 60x100 the heuristic would keep the patch kernel out
    out = {}
    try:
        for only in (True, False):
            # (PLANES_ONLY: the same elision for the layers on the plane kernels)
            eng = E.MadNetEngine(backend.lib, H, W, B=1, device=backend.device, weights=wn, precision="mixed", schedule=E.Schedule(SHADOW_ONLY=only, PLANES_ONLY=only))
            eng.set_inputs(l, r, gt[..., 0])
            plan = eng.build_plan("FULL", lr=1e-4, update=False)
            # (input gradients on mh_conv2d_planes_bwd, round 4: p = dz_hi, bank, mask_hi, dx, dx_hi -- fp32 elided = no dx, mask = a shadow's sign)
            from madnet_hip import _ffi
            pb = [plan.arr[k] for k in range(plan.n) if plan.arr[k].kind == _ffi.OP_CONV_PLANES_BWD]
            n_only = sum(1 for k in range(plan.n) if plan.arr[k].kind == 1 and plan.arr[k].i[23] & 4) + sum(1 for o in pb if not o.p[3])
            n_mask = sum(1 for k in range(plan.n) if plan.arr[k].kind == 1 and plan.arr[k].i[23] & 2) + sum(1 for o in pb if o.p[2])
            for k in E.LEVELS:
                for t in eng.dE[k]:
                    t.fill_(float("nan"))
            for t in eng.dCx:
                t.fill_(float("nan"))
            plan.run(backend.lib, 0)
            backend.sync()
            out[only] = (eng.params.g.clone(), n_only, n_mask)
    finally:
        if bname == "emul":
            backend.lib.tune_conv_patch(-1)
    (g1, n1, m1), (g0, n0, m0) = out[True], out[False]
    assert n0 == 0 and n1 >= 7 and m1 >= 9, (n0, n1, m1)
    assert torch.isfinite(g1).all() and ((g1 - g0).norm() / g0.norm()).item() <= 1e-6


@pytest.mark.parametrize("bname,size", [SIZES[0], pytest.param("hip", (375, 1242), marks=pytest.mark.gpu, id="hip-375x1242")])
def test_deterministic_mode_replays_bit_identical(bname, size):
    """MH_DETERMINISTIC (engine.DETERMINISTIC): the float atomics of the step -- bias gradients, the warp-gradient scatter --
    accumulate into 64-bit fixed-point twins (mh_deterministic_add) flushed behind every level's scatter and in front of the optimizer: two
    independent runs of the same two (MI355X: three) FULL steps give torch.equal weights and momentum (the emulator runs its workgroups on four threads, the
    MI355X on 256 CUs: the arrival order of the atomics differs from run to run), and agree with the default mode to its atomics noise."""
    backend = _backend(bname)
    H, W = size
    wn = S.calibrated_weights(OM.variable_shapes(), 1)
    l, r, gt = S.make_pair(H, W)
    res = []
    try:
        for det in (True, True, False):
            eng = E.MadNetEngine(backend.lib, H, W, B=1, device=backend.device, weights=wn, precision="mixed", schedule=E.Schedule(DETERMINISTIC=det))
            assert backend.lib.deterministic_ranges() == (2 if det else 0)
            eng.set_inputs(l, r, gt[..., 0])
            plan = eng.build_plan("FULL", lr=1e-3)
            plan.run(backend.lib, 0)
            backend.sync()
            first = (eng.params.w.clone(), eng.params.m.clone())
            for _ in range(2 if bname == "hip" else 1):
                plan.run(backend.lib, 0)
            backend.sync()
            res.append((eng.params.w.clone(), eng.params.m.clone(), eng.pred.clone(), first))
            eng.close()
            assert backend.lib.deterministic_ranges() == 0
    finally:
        pass
    (w0, m0, p0, f0), (w1, m1, p1, f1), (w2, m2, p2, f2) = res
    assert torch.equal(w0, w1) and torch.equal(m0, m1) and torch.equal(p0, p1)
    # against the default mode after ONE step (later steps amplify the atomics noise of the default mode through the adapted weights)
    scale = f2[1].abs().max().item()
    assert (f0[1] - f2[1]).abs().max().item() <= 1e-4 * scale and (f0[0] - f2[0]).abs().max().item() <= 1e-6


@pytest.mark.parametrize("net", ["madnet", "dispnet"])
@pytest.mark.parametrize("bname,size", [SIZES[0], pytest.param("hip", (375, 1242), marks=pytest.mark.gpu, id="hip-375x1242")])
def test_product_path_replays_bit_identical(bname, size, net):
    """The DEFAULT schedule (no MH_DETERMINISTIC twin) of the bench's arithmetic mode: two independent engines run the same three FULL adaptation steps from the
    same weights and end with torch.equal weights, momentum and disparity.  Round 6 took the last order-dependent arithmetic out of the
    step -- the bias gradients' one-atomic-per-workgroup sums became per-split partial sums reduced in split order by mh_wgrad_reduce (scripts/exp/det_probe.py
    found nothing else: filter gradients, feature gradients and disparity were already bit-identical from replay to replay).  The emulator runs its workgroups
    on four threads, the MI355X on 256 CUs: any arrival-order dependence shows up as a difference in the last bits."""
    backend = _backend(bname)
    H, W = size
    if net == "dispnet":
        from madnet_hip import dispnet_engine as DE
        from oracle import dispnet as OD
        if bname == "emul":
            H, W = 64, 128
        wn = S.calibrated_weights(OD.variable_shapes(), 1)
        mk = lambda: DE.DispNetEngine(backend.lib, H, W, B=1, device=backend.device, weights=wn, precision="mixed")
    else:
        wn = S.calibrated_weights(OM.variable_shapes(), 1)
        mk = lambda: E.MadNetEngine(backend.lib, H, W, B=1, device=backend.device, weights=wn, precision="mixed")
    pairs = [S.make_pair(H, W, frame=t) for t in range(3)]
    res = []
    for _ in range(2):
        eng = mk()
        assert not eng.deterministic and backend.lib.deterministic_ranges() == 0
        plan = eng.build_plan("FULL", lr=1e-3)
        for l, r, gt in pairs:
            eng.set_inputs(l, r, gt[..., 0])
            plan.run(backend.lib, 0)
        backend.sync()
        res.append((eng.params.w.clone(), eng.params.m.clone(), eng.params.g.clone(), eng.pred.clone()))
        if hasattr(eng, "close"):
            eng.close()
    for a, b in zip(res[0], res[1]):
        assert torch.isfinite(a).all() and torch.equal(a, b), (a - b).abs().max().item()
    assert (res[0][1] != 0).any()


@pytest.mark.gpu
@pytest.mark.parametrize("net", ["madnet", "dispnet"])
def test_captured_graph_replays_equal_the_eager_plan_bit_for_bit(hip, net):
    """The captured hipGraph of the default FULL step (side lanes as parallel branches, deferred side-lane launches, lane joins) against the SAME plan run eagerly on
    streams: four steps from the same weights end with torch.equal weights, momentum, gradients and disparity.  With no order-dependent arithmetic left in the step this is an
    exact check of the graph's dependency structure -- round 6 met a form (a lane-to-lane join op created at once, r06_experiments.txt #20) whose captured replays drifted
    8e-6 from the eager run while every tolerance-based test stayed green."""
    H, W = 375, 1242
    if net == "dispnet":
        from madnet_hip import dispnet_engine as DE
        from oracle import dispnet as OD
        wn = S.calibrated_weights(OD.variable_shapes(), 1)
        mk = lambda: DE.DispNetEngine(hip.lib, H, W, B=1, device=hip.device, weights=wn, precision="mixed")
    else:
        wn = S.calibrated_weights(OM.variable_shapes(), 1)
        mk = lambda: E.MadNetEngine(hip.lib, H, W, B=1, device=hip.device, weights=wn, precision="mixed")
    pairs = [S.make_pair(H, W, frame=t) for t in range(4)]
    res = []
    for graph in (False, True):
        eng = mk()
        plan = eng.build_plan("FULL", lr=1e-3)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            if graph:
                eng.set_inputs(*[pairs[0][0], pairs[0][1], pairs[0][2][..., 0]])
                w_keep, m_keep = eng.params.w.clone(), eng.params.m.clone()
                plan.capture(hip.lib, st.cuda_stream)          # (capturing launches nothing: the state is what it was)
                assert torch.equal(eng.params.w, w_keep) and torch.equal(eng.params.m, m_keep)
            for l, r, gt in pairs:
                eng.set_inputs(l, r, gt[..., 0])
                st.synchronize()
                plan.launch(hip.lib, st.cuda_stream)
                st.synchronize()
        res.append((eng.params.w.clone(), eng.params.m.clone(), eng.params.g.clone(), eng.pred.clone()))
        if hasattr(eng, "close"):
            eng.close()
    for a, b in zip(res[0], res[1]):
        assert torch.isfinite(a).all() and torch.equal(a, b), (a - b).abs().max().item()


@pytest.mark.gpu
def test_mixed_drift_against_the_fp32_engine_is_bounded():
    """the bench's arithmetic ('mixed': bf16 gradients) and the exact-fp32 engine adapt side by side on the same frame-shifted synthetic video
    from the same weights (bench.py's drift protocol: stream 200, 8 frames, lr 1e-4); after 10 steps their disparities differ by <= 2e-2 px (measured 1.4e-2;
    1e-3 is a SINGLE-step statement).  With the bias gradients summed in a fixed order the number is a property of the build, not of the box."""
    backend = _backend("hip")
    H, W = 375, 1242
    wn = S.calibrated_weights(OM.variable_shapes(), 1)
    pairs = [S.make_pair(H, W, stream_id=200, frame=t) for t in range(8)]
    ea = E.MadNetEngine(backend.lib, H, W, B=1, device="cuda", weights=wn, precision="mixed")
    eb = E.MadNetEngine(backend.lib, H, W, B=1, device="cuda", weights=wn, precision="fp32")
    pa, pb = ea.build_plan("FULL", lr=1e-4), eb.build_plan("FULL", lr=1e-4)
    d1 = None
    for k in range(1, 11):
        l, r, g = pairs[(k - 1) % 8]
        for e, p in ((ea, pa), (eb, pb)):
            e.set_inputs(l, r, g[..., 0])
            p.run(backend.lib, 0)
        if k == 1:
            backend.sync()
            d1 = (ea.pred - eb.pred).abs().mean().item()
    backend.sync()
    d10 = (ea.pred - eb.pred).abs().mean().item()
    assert d1 <= 1e-3 and d10 <= 2e-2, (d1, d10)


@pytest.mark.parametrize("bname", [pytest.param("emul", id="emul"), pytest.param("hip", marks=pytest.mark.gpu, id="hip")])
@pytest.mark.parametrize("mode", ["FULL", "MAD4", "NONE"])
def test_mixed_forward_from_planes_matches_fp32_operand_path(bname, mode):
    """'mixed' with the split-bf16 forward layers on mh_conv2d_planes (activations as hi / lo bf16 planes, fp32 copies elided where nothing reads
    them) against the same step on the fp32-operand bank kernels: the same three products per element in another summation order -- disparity, loss
    and post-step weights agree to fp32 round-off; the plan really took the planes kernel and really dropped fp32 stores.  (60x100: the small-layer
    threshold is lowered so that the 1/4-resolution layers count as 'large'.)"""
    backend = _backend(bname)
    lib, dev = backend.lib, backend.device
    wn = S.calibrated_weights(OM.variable_shapes(), 1)
    l, r, gt = S.make_pair(60, 100)
    res = {}
    lib.tune_conv_bank(128); lib.tune_conv_patch(128)
    try:
        for planes in (True, False, "unfused"):
            if planes == "unfused" and mode != "FULL":
                continue
            eng = E.MadNetEngine(lib, 60, 100, B=1, device=dev, weights=wn, precision="mixed", schedule=E.Schedule(USE_PLANES=bool(planes), FUSE_SPLITS=(planes != "unfused")))
            eng.bank_small_maxpix = 128
            eng.set_inputs(l, r, gt[..., 0])
            if mode == "FULL":
                plan = eng.build_plan("FULL", lr=1e-3)
            elif mode == "NONE":
                plan = eng.build_plan("NONE")
            else:
                blocks = json.load(open(os.path.join(PKG, "block_config", "MadNet_full.json")))
                lv = OM.layer_variables()
                plan = eng.build_plan("MAD", lr=1e-3, block_vars=sum([lv[n] for n in blocks[4]], []), block_level=E.LEVELS[4])
            lib.tune_conv_planes(0)
            plan.run(lib, 0)
            backend.sync()
            from madnet_hip import _ffi
            nplanes = lib.tune_conv_planes(0)
            pops = [plan.arr[i] for i in range(plan.n)]
            kinds = [o.kind for o in pops]
            elided = sum(1 for o in pops if o.kind == _ffi.OP_CONV_PLANES and not o.p[4])
            res[planes] = dict(pred=eng.pred.cpu().clone(), loss=eng.res_loss[0].item(), w={n: eng.params.tensor(n).cpu().clone() for n in wn},
                               nplanes=nplanes, nrec=kinds.count(_ffi.OP_CONV_PLANES), elided=elided, splits=kinds.count(_ffi.OP_PLANE_SPLIT),
                               nbwd=kinds.count(_ffi.OP_CONV_PLANES_BWD), bwd_elided=sum(1 for o in pops if o.kind == _ffi.OP_CONV_PLANES_BWD and not o.p[3]))
    finally:
        lib.tune_conv_bank(-1); lib.tune_conv_patch(-1)
    a, b = res[True], res[False]
    if "unfused" in res:        # split launches in front of every consumer instead: the same planes, bit for bit the same step
        u = res["unfused"]
        assert u["splits"] == 3 and u["nrec"] == 14 and u["nplanes"] == 14 + u["nbwd"]      # (conv6's input now comes as planes from conv5, a plane kernel: one split less than in round 5)
        assert torch.equal(u["pred"], a["pred"])               # (the post-step weights carry the landing order of the fp32 atomics: compared below, against `b`)
    # conv4, conv6, 5 of estimator 2, 6 of the context network + (round 6) the stride-2 conv5 on the stride-2 plane kernel (Schedule.PLANES_S2_FWD)
    assert a["nrec"] == 14 and a["nplanes"] == 14 + a["nbwd"] and b["nrec"] == 0 and b["nplanes"] == 0, (a["nrec"], a["nplanes"], b["nrec"])
    # the input gradients of those layers on the same kernel (one plane): 4 of estimator 2 + 5 of the context network + conv4 / conv6 in the FULL step,
    # most of their fp32 gradient maps never stored
    assert a["nbwd"] == {"FULL": 12, "MAD4": 10, "NONE": 0}[mode] and b["nbwd"] == 0, a["nbwd"]      # (+ the stride-2 input gradient of conv3; conv5 on the accumulating form is Schedule.PLANES_S2_ACC: measured slower, off)
    assert mode == "NONE" or a["bwd_elided"] >= 5, a["bwd_elided"]
    # one split launch left: the concat-split of the context network's input; the inputs of conv4 / conv6 and the estimator's concat buffer get their
    # planes from their producers' epilogues (engine.FUSE_SPLITS)
    assert a["splits"] == 1
    assert a["elided"] >= (9 if mode != "MAD4" else 4), a["elided"]
    d = (a["pred"] - b["pred"]).abs()
    # (fp32 summation order of 13 layers -- 32x32x16 against 16x16x32 MFMA steps -- amplified by the x20 disparity scales: measured 2.6e-4 / 4.2e-5)
    assert d.max().item() <= 1e-3 and d.mean().item() <= 1.5e-4, (d.max().item(), d.mean().item())
    assert abs(a["loss"] - b["loss"]) <= 2e-6
    for n in wn:
        step = (b["w"][n] - torch.from_numpy(wn[n])).abs().max().item()
        assert (a["w"][n] - b["w"][n]).abs().max().item() <= 2e-2 * step + 2e-7 * max(1.0, b["w"][n].abs().max().item()), n      # (+ an ulp or two of the weight itself)
