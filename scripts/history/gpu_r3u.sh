#!/bin/bash
# round 3, call U: level-3 forward layers on the small-layer bank kernel (split-bf16) instead of the 64x128 bank kernel (120 workgroups)
TAG=${1:-r3u}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
SWEEP="base:MH_X=0 small8k:MH_CONV_BANK_SMALL_MAXPIX=8192 base2:MH_X=0 small8k2:MH_CONV_BANK_SMALL_MAXPIX=8192" bash scripts/gpu_sweep.sh $TAG
MH_CONV_BANK_SMALL_MAXPIX=8192 timeout 300 python scripts/plan_table.py 2>&1 | grep -E "^ +(3[0-9]|4[0-9]) kind" 
timeout 300 python bench.py --no-cpu-baseline --steps 50 --repeats 3 2>$OUT/bench.err | tail -1 > $OUT/bench.json
MH_CONV_BANK_SMALL_MAXPIX=8192 timeout 300 python bench.py --steps 50 --repeats 3 --no-paths --no-roofline --no-step-surface 2>$OUT/bench8k.err | tail -1 > $OUT/bench8k.json
python - <<PY
import json
for n in ("bench", "bench8k"):
    j = json.load(open("$OUT/%s.json" % n)); print(n, j["ms_per_step"], j["value"], "epe", j.get("epe_vs_oracle"), j.get("within_tolerance"))
PY
