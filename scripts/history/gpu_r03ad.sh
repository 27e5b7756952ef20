#!/bin/bash
# round 2, call BD: remaining thresholds re-checked in the final step
TAG=${1:-r03ad}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
run base MH_X=1
run small2048 MH_CONV_BANK_SMALL_MAXPIX=2048
run small8192 MH_CONV_BANK_SMALL_MAXPIX=8192
run group8k MH_WGRAD_GROUP_MAXM=8192
run group32k MH_WGRAD_GROUP_MAXM=32768
run splits96 MH_WGRAD_MAXSPLITS=96
run splits48 MH_WGRAD_MAXSPLITS=48
run lanes2 MH_WGRAD_LANES=2
run base2 MH_X=1
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]])
    except Exception as ex: print(f, "ERR", ex)
PY
