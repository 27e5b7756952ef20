import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/real-time-self-adaptive-deep-stereo_amd')
import torch
from madnet_hip import engine as E, synthetic as S, _ffi
lib = _ffi.lib()
H, W = 375, 1242
wn = S.calibrated_weights(dict(E.madnet_manifest()), 1)
l, r, gt = S.make_pair(H, W)
ref = None
for name, cut, variant, graph in (("late", False, "", True), ("lane4 graph", True, "lane4", True), ("lane4 eager", True, "lane4", False), ("lane0 graph", True, "lane0", True), ("lane4_defer graph", True, "lane4_defer", True)):
    os.environ["MH_CUT_VARIANT"] = variant
    eng = E.MadNetEngine(lib, H, W, B=1, device="cuda", weights=wn, precision="mixed", schedule=E.Schedule(CUT_UPDATE=cut))
    eng.set_inputs(l, r, gt[..., 0])
    plan = eng.build_plan("FULL", lr=1e-4)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        plan.run(lib, st.cuda_stream); st.synchronize()
        w1 = eng.params.w.clone()
        if graph:
            plan.capture(lib, st.cuda_stream)
        for _ in range(3):
            plan.launch(lib, st.cuda_stream); st.synchronize()
        w4 = eng.params.w.clone()
        t0 = time.perf_counter()
        for _ in range(50):
            plan.launch(lib, st.cuda_stream); st.synchronize()
        dt = (time.perf_counter() - t0) / 50 * 1e3
    if ref is None: ref = (w1, w4)
    print("%-20s %.4f ms/step   |w - w_late| after 1 step %.3e, after 4 steps %.3e" % (name, dt, (w1 - ref[0]).abs().max().item(), (w4 - ref[1]).abs().max().item()))
