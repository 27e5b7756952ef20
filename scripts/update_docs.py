"""Regenerates the numbers block of README.md and DESIGN.md (between `<!-- BENCH:BEGIN ... -->` and `<!-- BENCH:END -->`) from the committed bench records of a round, so that
the documents cannot disagree with the line they cite:

    profiles/<ROUND>_bench_line_default.json          the driver-format line (what BENCH_rNN.json's `parsed` holds)
    profiles/<ROUND>_bench_line_default_detail.json   the complete record of the same run (bench_detail.json)
    profiles/<ROUND>_bench_line_{none,mad,dispnet_mixed,private4,batched4,shared_model_1gpu,mad_shared_1gpu}.json   the variants (optional)

usage: [ROUND=r06] python scripts/update_docs.py            (run after scripts/collect_profiles.sh)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("ROUND", "r06")


def load(name):
    p = os.path.join(ROOT, "profiles", "%s_bench_line_%s.json" % (ROUND, name))
    if not os.path.exists(p):
        return None
    try:
        return json.load(open(p))
    except ValueError:
        return None


def f(x, nd=3):
    return "n/a" if x is None else ("%.*f" % (nd, x))


def block():
    d = load("default")
    if d is None:
        raise SystemExit("profiles/%s_bench_line_default.json is missing (scripts/collect_profiles.sh first)" % ROUND)
    det = load("default_detail") or {}
    rf, rc = d.get("roofline", {}), d.get("roofline_corr", {})
    cfg = d.get("configs", {})
    rows = []
    add = rows.append
    add("| configuration (1242×375, `%s` arithmetic, hipGraph replay, one MI355X) | result |" % d["config"].get("precision", "mixed"))
    add("|---|---|")
    add("| **MADNet full-backprop adaptation** (BASELINE config 2, `value`: frames resident in HBM) | **%s ms/step = %s adapted pairs/s/GPU**, disparity %.2e px off the fp32 CPU oracle "
        "(tolerance 1e-3 px, single step) |" % (f(d["ms_per_step"]), f(d["value"], 1), d.get("epe_vs_oracle", float("nan"))))
    ss = d.get("step_surface", {})
    add("| the same loop as the reference counts FPS (`step_surface`: `Adapter.step`, new 8-bit frame pair uploaded + loss / EPE read back every step) | %s ms = **%s pairs/s** |"
        % (f(ss.get("ms_per_step")), f(ss.get("value"), 1)))
    for key, label in (("mad", "MADNet MAD modular adaptation, `MadNet_piramid_only.json`, through `Adapter.step` (config 3)"), ("dispnet", "DispNet full adaptation, 81-shift cost volume (config 4)"),
                       ("private4", "four streams with private models on one GPU (branches of one graph)"), ("batched4", "four streams sharing one model, batched")):
        c = cfg.get(key)
        if c and "value" in c:
            extra = ", %.2e px off its oracle" % c["epe_vs_oracle"] if c.get("epe_vs_oracle") is not None and key in ("mad", "dispnet") else ""
            add("| %s | %s ms = **%s pairs/s%s**%s |" % (label, f(c["ms_per_step"]), f(c["value"], 1), "/GPU" if key in ("private4", "batched4") else "", extra))
    n = load("none")
    if n:
        add("| MADNet forward only (`--mode NONE`: inference + loss + metrics) | %s ms = **%s pairs/s** |" % (f(n["ms_per_step"]), f(n["value"], 1)))
    for name, label in (("shared_model_1gpu", "FULL step with the shared-model collective INSIDE the graph (1-rank RCCL communicator: launch pattern, no wire)"),
                        ("mad_shared_1gpu", "MAD step, shared model, collective inside the graph (1 rank)")):
        s = load(name)
        if s:
            sm = s.get("shared_model") or {}
            add("| %s | %s ms/step%s |" % (label, f(s["ms_per_step"]), (" (collectives cost %s ms in the step)" % f(sm.get("collective_ms_in_step"))) if sm.get("collective_ms_in_step") is not None else ""))
    paths = d.get("paths", {})
    if paths:
        add("| the other arithmetic modes of the same step | " + "; ".join("`%s` %s ms = %s pairs/s, %.1e px" % (k, f(v.get("ms_per_step")), f(v.get("value"), 1), v.get("epe_vs_oracle", float("nan")))
                                                                         for k, v in paths.items()) + " |")
    dr = d.get("drift", {})
    if dr:
        add("| drift of `%s` against the exact-fp32 engine over consecutive adaptation steps (reported; 10 steps gated ≤ 2e-2 px) | " % d["config"].get("precision", "mixed") +
            ", ".join("%s: %.2e px" % (k.replace("step_", "step "), v) for k, v in sorted(dr.items(), key=lambda kv: int(kv[0].split("_")[1]) if kv[0].startswith("step_") else 0) if isinstance(v, float)) + " |")
    if rf:
        tr = rf.get("traffic")
        add("| dominant kernel family `%s` (%d launches, %s µs/step) | **%s of the dense bf16 MFMA peak** by algorithmic flops (%s by MFMA issue), HBM traffic %s |"
            % (rf.get("kernel"), int(rf.get("launches_per_step", 0)), f(rf.get("us_per_step"), 1), f(rf.get("frac")), f(rf.get("mfma_issue_frac")),
               ("%.2f× algorithmic (%.0f / %.0f MB)" % (tr / rf["algorithmic_bytes_per_step"], tr / 1e6, rf["algorithmic_bytes_per_step"] / 1e6)) if tr and rf.get("algorithmic_bytes_per_step") else "n/a"))
    if rc:
        def cf(k):
            e = rc.get(k, {})
            return "%s" % f(e.get("frac")) + ((" (traffic %.2f×)" % (e["traffic"] / e["algorithmic_bytes_per_launch"])) if e.get("traffic") and e.get("algorithmic_bytes_per_launch") else "")
        add("| correlation layer against the HBM roofline (8 TB/s; SURVEY 8(d) protocol shapes) | forward D = 5 %s, D = 81 %s; backward: fused level back end %s, plain D = 5 %s, D = 81 %s |"
            % (cf("fwd_d5"), cf("fwd_d81"), cf("warp_bwd_d5"), cf("bwd_d5"), cf("bwd_d81")))
    fam = det.get("kernel_families") or []
    if fam:
        add("| kernel families of the step (plan table, µs/step) | " + "; ".join("`%s` %s" % (x.get("kernel"), f(x.get("us_per_step"), 0)) for x in fam[:6]) +
            "; sum of all launches %s µs, replay / sum %s |" % (f(det.get("kernel_time_sum_us"), 0), f((det.get("box") or {}).get("replay_over_launch_sum"), 2)))
    cb = d.get("cpu_baseline", {})
    if cb:
        add("| CPU stand-in beside it (torch-CPU oracle, %d threads, kind `%s`; the reference's TF1 path cannot run here) | %s pairs/s |" % (cb.get("cores", 0), cb.get("kind"), f(cb.get("value"), 2)))
    src = "Source: `profiles/%s_bench_line_default.json` (+ `_detail`, variants `profiles/%s_bench_line_*.json`); the driver's own record of the round is `BENCH_%s.json`." % (ROUND, ROUND, ROUND.replace("r0", "r0"))
    return "\n".join(rows) + "\n\n" + src + "\n"


def main():
    b = block()
    for name in ("README.md", "DESIGN.md"):
        p = os.path.join(ROOT, name)
        s = open(p).read()
        m = re.search(r"(<!-- BENCH:BEGIN[^\n]*-->\n)(.*?)(<!-- BENCH:END -->)", s, re.S)
        if not m:
            print("%s: no BENCH block" % name)
            continue
        s = s[:m.start(2)] + b + s[m.start(3):]
        open(p, "w").write(s)
        print("%s: block regenerated (%d bytes)" % (name, len(b)))
    if "--print" in sys.argv:
        print(b)


if __name__ == "__main__":
    main()
