#!/bin/bash
# round 2, call Y: A/B of two bank-kernel thresholds (split-bf16 bank kernel for N = 32 layers; small-layer kernel for the 1/8-resolution input gradients)
TAG=${1:-r02y}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
run base MH_X=1
run minn32 MH_CONV_BANK_MIN_N=32
run dgrad8192 MH_CONV_BANK_SMALL_MAXPIX_DGRAD=8192
run both MH_CONV_BANK_MIN_N=32 MH_CONV_BANK_SMALL_MAXPIX_DGRAD=8192
run base_again MH_X=1
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e["config"].get("ops_per_step"))
PY
