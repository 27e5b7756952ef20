#!/bin/bash
# round 3, call P: row-streaming kernel with the loads two rows ahead
TAG=${1:-r3p}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_parity.py -m gpu -x -q -k "conv_rows" 2>&1 | tail -3
SWEEP="base:MH_X=0 rowsoff:MH_CONV_ROWS_MINPIX=0 rows32k:MH_CONV_ROWS_MINPIX=32768 base2:MH_X=0" bash scripts/gpu_sweep.sh $TAG
timeout 300 python scripts/plan_table.py > $OUT/plan_table_madnet.txt 2>&1; grep -E "conv_rows|ops," $OUT/plan_table_madnet.txt | head
MH_CONV_ROWS_MINPIX=32768 timeout 300 python scripts/plan_table.py 2>&1 | grep -E "^ +[0-9]+ kind.*conv_rows" | sed "s/^/32k /"
