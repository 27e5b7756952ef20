#!/bin/bash
# round 2, call AB: grouped filter gradients for ALL layers now that the side lane mostly runs after the input-gradient chain
TAG=${1:-r03b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
run m16k MH_X=1
run m32k MH_WGRAD_GROUP_MAXM=32768
run m64k MH_WGRAD_GROUP_MAXM=65536
run mall MH_WGRAD_GROUP_MAXM=100000000
run mall_splits MH_WGRAD_GROUP_MAXM=100000000 MH_WGRAD_MAXSPLITS=32
run m16k_splits64 MH_WGRAD_MAXSPLITS=64
EXTRA="--mode MAD" run mad_m16k MH_X=1
EXTRA="--mode MAD" run mad_mall MH_WGRAD_GROUP_MAXM=100000000
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e["config"].get("ops_per_step"))
PY
