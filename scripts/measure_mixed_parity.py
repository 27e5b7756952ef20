"""Measured deviations of the 'mixed' mode from the fp32 CPU oracle at the headline shape (375x1242): FULL step, MAD blocks 0 and 4, DispNet FULL.
The parity tests assert 4x these values (tests/test_engine_parity.py, tests/test_dispnet_parity.py)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")):
    sys.path.insert(0, p)
import torch
from madnet_hip import _ffi, engine as E, dispnet_engine as DE, synthetic as S
from oracle import madnet as OM, dispnet as OD

lib = _ffi.lib()
H, W = 375, 1242
PKG = os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd")


def metrics(eng, wn, wt, o, names=None):
    torch.cuda.synchronize()
    epe = (eng.pred.cpu() - o["disparity"][..., 0]).abs().mean().item()
    gh = torch.cat([eng.params.tensor(n, "g").cpu().flatten() for n in o["grads"]])
    go = torch.cat([g.flatten() for g in o["grads"].values()])
    cos = torch.nn.functional.cosine_similarity(gh, go, dim=0).item()
    grel = (gh - go).norm().item() / go.norm().item()
    per = {n: (eng.params.tensor(n, "g").cpu() - g).norm().item() / max(g.norm().item(), 1e-30) for n, g in o["grads"].items()}
    worst = max(per, key=per.get)
    dw = max((eng.params.tensor(n).cpu() - wt[n]).abs().max().item() for n in o["grads"])
    step = max((torch.from_numpy(wn[n]) - wt[n]).abs().max().item() for n in o["grads"])
    untouched = all(torch.equal(eng.params.tensor(n).cpu(), torch.from_numpy(wn[n])) for n in wt if n not in o["grads"])
    return dict(epe=epe, cos=cos, grel=grel, worst_tensor=worst, worst_rel=per[worst], dw=dw, step=step, dw_over_step=dw / step,
                loss_err=abs(eng.res_loss[0].item() - o["loss"]), untouched_bit_identical=untouched)


wn = S.calibrated_weights(OM.variable_shapes(), 1)
l, r, gt = S.make_pair(H, W)
tl, tr, tg = (torch.from_numpy(a) for a in (l, r, gt))
out = {}
for tag, mode, block in (("FULL", "FULL", None), ("MAD0", "MAD", 0), ("MAD4", "MAD", 4)):
    eng = E.MadNetEngine(lib, H, W, B=1, device="cuda", weights=wn, precision="mixed")
    eng.set_inputs(l, r, gt[..., 0])
    wt = {k: torch.from_numpy(v.copy()) for k, v in wn.items()}
    acc = {k: torch.zeros_like(v) for k, v in wt.items()}
    lr = 1e-4
    if mode == "FULL":
        eng.build_plan("FULL", lr=lr).run(lib, 0)
        o = OM.step(wt, acc, tl, tr, tg, mode="FULL", lr=lr)
    else:
        blocks = json.load(open(os.path.join(PKG, "block_config", "MadNet_full.json")))
        lv = OM.layer_variables()
        bv = sum([lv[n] for n in blocks[block]], [])
        eng.build_plan("MAD", lr=lr, block_vars=bv, block_level=E.LEVELS[block]).run(lib, 0)
        o = OM.step(wt, acc, tl, tr, tg, mode="MAD", block_vars=bv, block_index=block, lr=lr)
    out[tag] = metrics(eng, wn, wt, o)
    print(tag, out[tag], flush=True)
wn = S.calibrated_weights(OD.variable_shapes(), 1)
eng = DE.DispNetEngine(lib, H, W, B=1, device="cuda", weights=wn, precision="mixed")
eng.set_inputs(l, r, gt[..., 0])
wt = {k: torch.from_numpy(v.copy()) for k, v in wn.items()}
acc = {k: torch.zeros_like(v) for k, v in wt.items()}
eng.build_plan("FULL", lr=1e-4).run(lib, 0)
o = OD.step(wt, acc, tl, tr, tg, mode="FULL", lr=1e-4)
out["DISPNET_FULL"] = metrics(eng, wn, wt, o)
print("DISPNET_FULL", out["DISPNET_FULL"], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "mixed_parity.json"), "w"), indent=1)
