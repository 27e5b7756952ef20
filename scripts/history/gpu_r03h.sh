#!/bin/bash
# round 2, call AH: 8-wave 128x128 filter-gradient tile, same-box A/B
TAG=${1:-r03h}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
run w8_1 MH_WGRAD_W8=1
run w8_0 MH_WGRAD_W8=0
run w8_1b MH_WGRAD_W8=1
run w8_0b MH_WGRAD_W8=0
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e["config"].get("ops_per_step"))
PY
