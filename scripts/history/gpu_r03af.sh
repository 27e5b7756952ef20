#!/bin/bash
# round 2, call BF: per-plan grouped filter-gradient cap (DispNet 16384, MADNet 4096) + round-1 split targets above 65536 reduction pixels
TAG=${1:-r03af}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
EXTRA="--steps 100" run full MH_X=1
EXTRA="--steps 30 --streams-per-gpu 4" run batched4 MH_X=1
EXTRA="--steps 20 --streams-per-gpu 8" run batched8 MH_X=1
EXTRA="--steps 30 --model dispnet" run dispnet_mixed MH_X=1
EXTRA="--steps 30 --model dispnet" run dispnet_mixed_g4k MH_DISPNET_GROUP_MAXM=4096
EXTRA="--steps 30 --model dispnet --precision bf16" run dispnet_bf16 MH_X=1
EXTRA="--steps 100 --mode MAD" run mad MH_X=1
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        e=json.load(open(f)); print(f.split("/")[-1], "%.1f pairs/s"%e["value"], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]])
    except Exception as ex: print(f, "ERR", ex)
PY
