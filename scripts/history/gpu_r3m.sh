#!/bin/bash
# round 3, call M: row-streaming kernel of the 16 -> 16 layer at 1/2 resolution (conv_rows.hip): parity on the GPU, A/B in the step, plan table
TAG=${1:-r3m}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_parity.py -m gpu -x -q -k "conv_rows or k1_dgrad" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_engine_parity.py -m gpu -x -q 2>&1 | tail -3
SWEEP="base:MH_X=0 rowsoff:MH_CONV_ROWS_MINPIX=0 base2:MH_X=0 rowsoff2:MH_CONV_ROWS_MINPIX=0" bash scripts/gpu_sweep.sh $TAG
timeout 300 python scripts/plan_table.py > $OUT/plan_table_madnet.txt 2>&1; grep -E "conv_rows|ops," $OUT/plan_table_madnet.txt | head
timeout 300 python bench.py --no-cpu-baseline --steps 50 --repeats 3 2>$OUT/bench.err | tail -1 > $OUT/bench.json
python - <<PY
import json
j = json.load(open("$OUT/bench.json")); print("bench", j["ms_per_step"], j["value"], "epe", j.get("epe_vs_oracle"), j.get("within_tolerance"))
PY
