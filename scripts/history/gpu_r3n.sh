#!/bin/bash
# round 3, call N: stride-2 instances of the row-streaming kernel (conv1, conv3 forward), rows-per-wave A/B
TAG=${1:-r3n}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_parity.py -m gpu -x -q -k "conv_rows" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_engine_parity.py -m gpu -x -q 2>&1 | tail -3
SWEEP="base:MH_X=0 r6:MH_CONV_ROWS_R=6 r8:MH_CONV_ROWS_R=8 r3:MH_CONV_ROWS_R=3 rowsoff:MH_CONV_ROWS_MINPIX=0 base2:MH_X=0" bash scripts/gpu_sweep.sh $TAG
timeout 300 python scripts/plan_table.py > $OUT/plan_table_madnet.txt 2>&1; grep -E "conv_rows|ops," $OUT/plan_table_madnet.txt | head
for r in 3 6 8; do MH_CONV_ROWS_R=$r timeout 300 python scripts/plan_table.py 2>&1 | grep -E "^ +[0-9]+ kind.*conv_rows" | sed "s/^/R=$r /"; done
timeout 300 python bench.py --no-cpu-baseline --steps 50 --repeats 3 2>$OUT/bench.err | tail -1 > $OUT/bench.json
python - <<PY
import json
j = json.load(open("$OUT/bench.json")); print("bench", j["ms_per_step"], j["value"], "epe", j.get("epe_vs_oracle"), j.get("within_tolerance"))
PY
