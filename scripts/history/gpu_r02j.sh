#!/bin/bash
TAG=${1:-r02j}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
run a_default MH_X=1
run b_noscatter MH_SCATTER_LANE=0
run c_noscatter_nosideloss MH_SCATTER_LANE=0 MH_SIDE_LOSS=0
run d_scatter2 MH_SCATTER_LANE=2
run e_scatter1 MH_SCATTER_LANE=1
run f_scatter4 MH_SCATTER_LANE=4
run g_default_again MH_X=1
run h_nosideloss MH_SIDE_LOSS=0
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e["config"].get("ops_per_step"))
PY
