#!/bin/bash
# round 2, call BB: filter-gradient workgroup targets (pixel splits) in the deferred one-lane step
TAG=${1:-r03ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
run wg100 MH_WGRAD_TARGET_PCT=100
run wg66 MH_WGRAD_TARGET_PCT=66
run wg50 MH_WGRAD_TARGET_PCT=50
run wg40 MH_WGRAD_TARGET_PCT=40
run wg33 MH_WGRAD_TARGET_PCT=33
run wg25 MH_WGRAD_TARGET_PCT=25
EXTRA="--mode MAD" run mad_wg50 MH_WGRAD_TARGET_PCT=50
EXTRA="--mode MAD" run mad_wg100 MH_WGRAD_TARGET_PCT=100
EXTRA="--precision bf16" run bf16_wg50 MH_WGRAD_TARGET_PCT=50
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e["step_aggregate"].get("wgrad_ws_bytes") if "step_aggregate" in e else None)
    except Exception as ex: print(f, "ERR", ex)
PY
