#!/bin/bash
# round 2, call BE: grouped filter-gradient launches: pixel cap
TAG=${1:-r03ae}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
run g16k MH_WGRAD_GROUP_MAXM=16384
run g8k MH_WGRAD_GROUP_MAXM=8192
run g4k MH_WGRAD_GROUP_MAXM=4096
run g2k MH_WGRAD_GROUP_MAXM=2048
run g8k_b MH_WGRAD_GROUP_MAXM=8192
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]])
PY
