"""Summarise the rocprofv3 --pmc passes of scripts/gpu_pmc_r04.sh into profiles/r04_pmc_roofline.json: one entry per kernel string of the recorded plan
(what mh_last_kernel reports = what bench.py prints), plus the fixed roofline kernels.  usage: python scripts/pmc_summarize_r04.py gpurun_out/<tag>"""
import csv, glob, json, os, sys, collections

src = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
meta = json.load(open(os.path.join(src, "ops.json")))
REPS = meta["reps"]


def dispatches(d):
    """[(dispatch id order) -> {kernel, counters{}, dur_us}] of one pass, in launch order"""
    fs = glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        return None
    by = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        k = int(r["Dispatch_Id"])
        e = by.setdefault(k, {"kernel": r["Kernel_Name"], "grid": r.get("Grid_Size", ""), "c": {}, "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 if "End_Timestamp" in r else None})
        e["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    return [by[k] for k in sorted(by)]


def split_ops(disp):
    """groups of dispatches between the fill separators: [group] for the plan part, then the tail (fixed roofline kernels)"""
    groups, cur, tail, nsep = [], None, None, 0
    for k, e in enumerate(disp):
        if "fill_kernel" in e["kernel"]:
            nsep += 1
            if cur is not None:
                groups.append(cur)
            cur = []
            if nsep >= 2 and k + 1 < len(disp) and "fill_kernel" in disp[k - 1]["kernel"]:
                tail = disp[k + 1:]
                cur = None
                break
            continue
        if cur is not None:
            cur.append(e)
    return [g for g in groups if g], tail or []


out = {"source": "scripts/gpu_pmc_r04.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | two SQ_* sets in separate passes (--kernel-trace only) over scripts/pmc_plan_r04.py: every "
                 "conv / filter-gradient / correlation op of the recorded MADNet FULL plan ('mixed', 1242x375) launched alone %d times, keyed by the kernel string bench.py "
                 "reports.  FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B); "
                 "traffic_bytes = 2 * FETCH + WRITE of ONE launch behind a 96 MB fill (L2 flushed: what a layer sees in the step, where its input has just been written back by the previous kernel), summed over the kernels the op launches" % REPS}
passes = {d: dispatches(d) for d in ("FETCH_SIZE", "WRITE_SIZE", "SQ", "SQ2")}
per_op = {}
for d, disp in passes.items():
    if disp is None:
        continue
    groups, tail = split_ops(disp)
    groups = groups[-len(meta["ops"]):]        # (the plan's own zero fills in the warm-up run open spurious groups in front of the first separator)
    if len(groups) != len(meta["ops"]):
        print("WARNING: pass %s has %d op groups, the driver launched %d" % (d, len(groups), len(meta["ops"])))
    for g, op in zip(groups, meta["ops"]):
        n = max(1, len(g) // REPS)
        last = g[-n:]                                          # the kernels of the op's last launch
        e = per_op.setdefault(op["kernel"], {"plan_op_index": op["index"], "algorithmic_flops": op["flops"], "algorithmic_bytes": op["bytes"], "kernels_per_launch": n})
        for c in last[0]["c"]:
            e[c] = sum(x["c"].get(c, 0.0) for x in last)
        if last[0]["us"] is not None:
            e["launch_us_under_pmc"] = round(sum(x["us"] for x in last), 1)
    # fixed kernels: the tail is a sequence of runs of one kernel each, in the order benchtools.roofline launches them
    runs = []
    for e in tail:
        if not any(t in e["kernel"] for t in ("conv_planes_kernel", "conv_bank_kernel", "conv_patch_kernel", "conv_igemm_kernel", "wgrad_stream_kernel", "wgrad_bf16_kernel", "corr_fwd")):
            continue                                            # (casts, table uploads, torch fills)
        sig = (e["kernel"], e["grid"])
        if not runs or runs[-1][0] != sig:
            runs.append((sig, []))
        runs[-1][1].append(e)
    order = ["roofline_fwd", "roofline_dgrad", "roofline_wgrad", "roofline_wgrad_batch", "roofline_corr", "roofline_corr_b1"]
    for (sig, es), name in zip(runs, order):
        kstr = meta.get("fixed", {}).get(name)
        if not kstr:
            continue
        ee = per_op.setdefault(kstr, {"fixed_roofline_entry": name, "rocprof_kernel": sig[0][:100]})
        for c in es[0]["c"]:
            v = sorted(x["c"].get(c, 0.0) for x in es)
            ee[c] = v[len(v) // 2]
for k, e in per_op.items():
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["traffic_bytes"] = int(2 * e["FETCH_SIZE"] * 1024 + e["WRITE_SIZE"] * 1024)
    out[k] = e
# aliases for the fixed roofline entries of bench.py (same kernel strings where the plan launches the same shape)
out["fixed_kernels"] = meta.get("fixed", {})
json.dump(out, open(os.path.join(ROOT, "profiles", "r04_pmc_roofline.json"), "w"), indent=1)
print("%d keys" % len(out))
for k, e in list(out.items())[:400]:
    if isinstance(e, dict) and "traffic_bytes" in e:
        print("%-110s traffic %7.2f MB  alg %7.2f MB" % (k[:110], e["traffic_bytes"] / 1e6, e.get("algorithmic_bytes", 0) / 1e6))
