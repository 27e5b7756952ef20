"""Summarise the rocprofv3 --pmc passes of scripts/gpu_pmc_r02.sh into profiles/r02_pmc_roofline.json (+ raw CSVs in profiles/r02_pmc/).
usage: python scripts/pmc_summarize_r02.py gpurun_out/<tag>"""
import csv, glob, json, os, shutil, sys

src = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles", "r02_pmc")
os.makedirs(dst, exist_ok=True)


def load(d):
    fs = glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        return None
    shutil.copy(fs[0], os.path.join(dst, d + "_counter_collection.csv"))
    return list(csv.DictReader(open(fs[0])))


def mean_of(rows, pred, counter):
    sel = [r for r in rows if r["Counter_Name"] == counter and pred(r["Kernel_Name"])]
    v = [float(r["Counter_Value"]) for r in sel]
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel]
    v, d = v[1:] or v, d[1:] or d                     # drop the first (cold) launch
    return (sum(v) / len(v), sum(d) / len(d) / 1e3) if v else (None, None)


CONV_BYTES = 2 * 15728640 + 589824 + 512
KERNELS = {
    "conv_fwd_x3_bank_3x3_128_128_96x320": (lambda n: "conv_bank_kernel<2, 4, 2, 2, true>" in n, CONV_BYTES, 9059696640.0),
    "conv_fwd_bf16_bank_small_3x3_128_128_24x80": (lambda n: "conv_bank_small_kernel<false, 1>" in n, 2 * 24 * 80 * 128 * 4 + 589824 // 2 + 512, 2.0 * 24 * 80 * 9 * 128 * 128),
    "conv_fwd_x3_patch_3x3_128_128_96x320": (lambda n: "conv_patch_kernel" in n and "false, 4, true" in n, CONV_BYTES, 9059696640.0),
    "conv_fwd_bf16_patch_3x3_128_128_96x320": (lambda n: "conv_patch_kernel" in n and "false, 4, false" in n, CONV_BYTES, 9059696640.0),
    "conv_dgrad_bf16_patch_3x3_128_128_96x320": (lambda n: "conv_patch_kernel" in n and "true, 4, false" in n, CONV_BYTES + 15728640, 9059696640.0),
    "wgrad_bf16_partial_3x3_128_128_96x320": (lambda n: "wgrad_bf16_kernel<2, 2, 4, 4>" in n, CONV_BYTES, 9059696640.0),
    "corr_fwd_B64_96x320x32_D5": (lambda n: "corr_fwd_direct" in n, 64 * 96 * 320 * (2 * 32 + 5) * 4, None),
    "corr_fwd_bf16_mfma_B16_96x320x128_D81": (lambda n: "corr_fwd_mfma_bf16" in n, 16 * 96 * 320 * (2 * 128 + 81) * 4, None),
}
fetch, write, sq, sq2 = load("FETCH_SIZE"), load("WRITE_SIZE"), load("SQ"), load("SQ2")
out = {"source": "scripts/gpu_pmc_r02.sh (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / two SQ_* sets in separate passes, --kernel-trace only), raw CSVs "
                 "in profiles/r02_pmc/; FETCH_SIZE and WRITE_SIZE are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies the 128-B requests "
                 "of wide coalesced reads at 64 B); mean over launches 2..5 of scripts/pmc_kernels_r02.py"}
for key, (pred, alg, flops) in KERNELS.items():
    e = {"algorithmic_bytes": alg}
    if fetch is not None and write is not None:
        f, us = mean_of(fetch, pred, "FETCH_SIZE")
        w, _ = mean_of(write, pred, "WRITE_SIZE")
        if f is not None and w is not None:
            e.update({"fetch_kib_raw": round(f, 1), "write_kib": round(w, 1), "traffic_bytes": int(2 * f * 1024 + w * 1024), "launch_us_under_pmc": round(us, 1)})
    for rows in (sq, sq2):
        if rows is None:
            continue
        for c in sorted(set(r["Counter_Name"] for r in rows)):
            v, _ = mean_of(rows, pred, c)
            if v is not None:
                e[c] = int(v)
    if flops:
        e["algorithmic_flops"] = flops
    out[key] = e
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_pmc_roofline.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
