#!/bin/bash
OUT=gpurun_out/r3i; mkdir -p $OUT
( time timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/time.txt; tail -3 $OUT/time.txt; tail -5 $OUT/bench.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r3i/bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "epe_vs_oracle", "within_tolerance"): print(k, j.get(k))
print("timing", j["timing"])
for k in ("roofline", "roofline_fwd", "roofline_dgrad", "roofline_wgrad", "roofline_wgrad_batch", "roofline_corr"):
    r = j.get(k, {}); print(k, {x: r.get(x) for x in ("kernel", "achieved", "frac", "mfma_issue_frac", "launch_ms", "traffic", "family_us_per_step", "family_share_of_kernel_time", "error")})
print("families", j.get("kernel_families"))
print("drift", j.get("drift"))
print("step_surface", j.get("step_surface")); print("cpu", j.get("cpu_baseline")); print("paths", j.get("paths")); print("agg", j.get("step_aggregate"))
PY
bash scripts/gpu_pmc_r03.sh r3pmc 2>&1 | tail -70
