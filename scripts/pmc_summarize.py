"""Summarise the rocprofv3 --pmc passes of scripts/gpu_pmc.sh into profiles/<ROUND>_pmc_roofline.json (ROUND from the environment, default r06): one entry per kernel string of the recorded plans
(MADNet FULL, the MAD block plans, DispNet FULL: what mh_last_kernel reports = what bench.py prints) plus the fixed roofline entries.
usage: [ROUND=r06] python scripts/pmc_summarize.py gpurun_out/<tag>"""
import collections
import csv
import glob
import json
import os
import sys

src = sys.argv[1]
ROUND = os.environ.get("ROUND", "r06")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
meta = json.load(open(os.path.join(src, "ops.json")))
G = meta["groups"]


def dispatches(d):
    fs = glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        return None
    by = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        k = int(r["Dispatch_Id"])
        e = by.setdefault(k, {"kernel": r["Kernel_Name"], "grid": r.get("Grid_Size", ""), "c": {},
                              "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 if "End_Timestamp" in r else None})
        e["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    return [by[k] for k in sorted(by)]


def split(disp):
    """dispatch groups between the 96 MB separators (fill_kernel with the separator's grid), and the tail behind the double separator"""
    sizes = collections.Counter(e["grid"] for e in disp if "fill_kernel" in e["kernel"])
    sep_grid = max(sizes, key=lambda k: (int(k) if str(k).isdigit() else 0))          # the separator is the largest fill of the run
    groups, cur, tail = [], None, []
    for k, e in enumerate(disp):
        is_sep = "fill_kernel" in e["kernel"] and e["grid"] == sep_grid
        if is_sep:
            if cur is not None:
                groups.append(cur)
            cur = []
            if k > 0 and "fill_kernel" in disp[k - 1]["kernel"] and disp[k - 1]["grid"] == sep_grid:
                tail = disp[k + 1:]
                cur = None
                groups = groups[:-1] if groups and not groups[-1] else groups
                break
            continue
        if cur is not None:
            cur.append(e)
    if cur:
        groups.append(cur)
    return [g for g in groups if g], tail


out = {"source": "scripts/gpu_pmc.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_* in separate passes (--kernel-trace only) over scripts/pmc_plan.py: every conv / "
                 "filter-gradient / correlation op of the recorded MADNet FULL plan, the MAD block plans and the DispNet FULL plan ('mixed', 1242x375) launched alone, keyed by the "
                 "kernel string bench.py reports, plus the fixed roofline entries.  FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies the "
                 "128-B requests of wide coalesced reads at 64 B); traffic_bytes = 2 * FETCH + WRITE of ONE launch behind a 96 MB fill (L2 flushed: what a layer sees in the step, where "
                 "its input has just been written back by the previous kernel), summed over the kernels the op launches"}
per = {}
for d in ("FETCH_SIZE", "WRITE_SIZE", "SQ"):
    disp = dispatches(d)
    if disp is None:
        continue
    groups, tail = split(disp)
    groups = groups[-len(G):]
    if len(groups) != len(G):
        print("WARNING: pass %s has %d groups, the driver launched %d" % (d, len(groups), len(G)))
    for g, op in zip(groups, G):
        if op.get("dup"):
            continue
        key = op["kernel"] if not op.get("fixed") else "%s [%s]" % (op["kernel"], op["fixed"])       # (a fixed entry never shares its counters with a plan op of the same kernel string)
        e = per.setdefault(key, {"plan": op["plan"], "plan_op_index": op["index"], "algorithmic_flops": op["flops"], "algorithmic_bytes": op["bytes"], "kernels_per_launch": len(g)})
        if op.get("fixed"):
            e["fixed_roofline_entry"] = op["fixed"]
            meta["fixed"][op["fixed"]] = key
        for c in g[0]["c"]:
            e[c] = sum(x["c"].get(c, 0.0) for x in g)
        if g[0]["us"] is not None:
            e["launch_us_under_pmc"] = round(sum(x["us"] for x in g), 1)
    runs = []
    for e in tail:
        if not any(t in e["kernel"] for t in ("conv_planes_kernel", "conv_bank_kernel", "conv_patch_kernel", "conv_igemm_kernel", "wgrad_stream_kernel", "wgrad_bf16_kernel", "corr_fwd")):
            continue
        sig = (e["kernel"], e["grid"])
        if not runs or runs[-1][0] != sig:
            runs.append((sig, []))
        runs[-1][1].append(e)
    order = ["roofline_fwd", "roofline_dgrad", "roofline_wgrad", "roofline_wgrad_batch", "roofline_corr", "roofline_corr_b1"]
    for (sig, es), name in zip(runs, order):
        kstr = meta.get("fixed", {}).get(name)
        if not kstr or kstr in per and "plan_op_index" in per[kstr] and per[kstr].get("plan") != "tail":
            continue                                        # (the same kernel string was measured as a plan op: that entry stands)
        ee = per.setdefault(kstr, {"plan": "tail", "fixed_roofline_entry": name, "rocprof_kernel": sig[0][:100]})
        for c in es[0]["c"]:
            v = sorted(x["c"].get(c, 0.0) for x in es)
            ee[c] = v[len(v) // 2]
for k, e in per.items():
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["traffic_bytes"] = int(2 * e["FETCH_SIZE"] * 1024 + e["WRITE_SIZE"] * 1024)
    out[k] = e
out["fixed_kernels"] = meta.get("fixed", {})
if os.environ.get("PMC_TUNE"):
    out["tuning_hooks"] = os.environ["PMC_TUNE"]
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_pmc_roofline%s.json" % (ROUND, os.environ.get("PMC_SUFFIX", ""))), "w"), indent=1)
print("%d keys" % len(out))
for k, e in out.items():
    if isinstance(e, dict) and "traffic_bytes" in e:
        print("%-118s %-14s traffic %8.2f MB  alg %8.2f MB" % (k[:118], e.get("plan", ""), e["traffic_bytes"] / 1e6, e.get("algorithmic_bytes", 0) / 1e6))
