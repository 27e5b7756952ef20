import faulthandler, sys, os
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-time-self-adaptive-deep-stereo_amd"))
import torch, ctypes as C
from madnet_hip import _ffi, engine as E, synthetic as S
from madnet_hip.plan import MultiPlan
lib = _ffi.lib()
H, W = 128, 256
shapes = dict(E.madnet_manifest())
engs = []
for i in range(2):
    wn = S.calibrated_weights(shapes, 1 + i)
    l, r, gt = S.make_pair(H, W, stream_id=i)
    e = E.MadNetEngine(lib, H, W, B=1, device="cuda", weights=wn, precision="mixed")
    e.set_inputs(l, r, gt[..., 0]); engs.append(e)
plans = [e.build_plan("FULL", lr=1e-3) for e in engs]
print("plans built", [p.n for p in plans], flush=True)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for p in plans:
        p.run(lib, st.cuda_stream)
    st.synchronize(); print("single runs ok", flush=True)
    mp = MultiPlan(plans)
    lib.plans_prepare(2); print("prepared", flush=True)
    lib.plans_run(mp.refs, 1, C.c_void_p(st.cuda_stream)); st.synchronize(); print("plans_run n=1 ok", flush=True)
    lib.plans_run(mp.refs, 2, C.c_void_p(st.cuda_stream)); st.synchronize(); print("plans_run n=2 eager ok", flush=True)
    def cap(refs, n, tag):
        s_ = C.c_void_p(st.cuda_stream)
        lib.graph_begin(s_); print(tag, "begin", flush=True)
        lib.plans_run(refs, n, s_); print(tag, "recorded", flush=True)
        g = C.c_void_p(); lib.graph_end(s_, C.byref(g)); print(tag, "instantiated", flush=True)
        for _ in range(3):
            lib.graph_launch(g, s_)
        st.synchronize(); print(tag, "replayed", flush=True)
    cap(mp.refs, 1, "n=1")
    which = sys.argv[1] if len(sys.argv) > 1 else "lanes"
    if which == "nolanes":
        for e in engs:
            e.wgrad_lanes = 0
        plans2 = [e.build_plan("FULL", lr=1e-3) for e in engs]
        mp2 = MultiPlan(plans2)
        cap(mp2.refs, 2, "n=2 no side lanes")
    else:
        cap(mp.refs, 2, "n=2")
