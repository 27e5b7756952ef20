#!/bin/bash
# round 2, call BA: more round-1 heuristics re-swept in the round-2 step (filter-gradient workgroup targets, flat mode, patch tile of the bf16 input gradients)
TAG=${1:-r03aa}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
run base MH_X=1
run wg66 MH_WGRAD_TARGET_PCT=66
run wg150 MH_WGRAD_TARGET_PCT=150
run wg200 MH_WGRAD_TARGET_PCT=200
run flat2 MH_WGRAD_FLAT=2
run flat0 MH_WGRAD_FLAT=0
run patch64 MH_CONV_PATCH=64
run patch128w4 MH_CONV_PATCH=128
run base2 MH_X=1
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]])
    except Exception as ex: print(f, "ERR", ex)
PY
