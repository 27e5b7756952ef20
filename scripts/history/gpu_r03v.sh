#!/bin/bash
# round 2, call AV: bank kernel for the <= 64-column split-bf16 layers at 1/8 resolution (A/B)
TAG=${1:-r03v}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
run min7680 MH_CONV_BANK_MINPIX=7680
run min24576 MH_CONV_BANK_MINPIX=24576
run min15360 MH_CONV_BANK_MINPIX=15360
run min7680b MH_CONV_BANK_MINPIX=7680
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]])
PY
