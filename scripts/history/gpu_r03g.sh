#!/bin/bash
# round 2, call AG: 8-wave 128x128 filter-gradient tile in situ + parity
TAG=${1:-r03g}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
timeout 600 python -m pytest tests/test_conv_parity.py tests/test_engine_parity.py -m gpu -x -q 2>&1 | tail -2
run w8 MH_X=1
EXTRA="--mode MAD" run mad_w8 MH_X=1
EXTRA="--precision bf16" run bf16_w8 MH_X=1
run w8_again MH_X=1
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e["config"].get("ops_per_step"))
PY
