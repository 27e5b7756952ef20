#!/bin/bash
# round 2, call AC: the per-step mh_pack_weights launch on the side lane
TAG=${1:-r03c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
run pack0 MH_PACK_LANE=0
run pack1 MH_PACK_LANE=1
run pack0_again MH_PACK_LANE=0
run pack1_again MH_PACK_LANE=1
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e["config"].get("ops_per_step"))
PY
