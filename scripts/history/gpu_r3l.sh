#!/bin/bash
# round 3, call L: one-launch loss tile kernel, row-scatter correlation/warp gradient (A/B of its row threshold)
TAG=${1:-r3l}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_parity.py tests/test_golden_kats.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_engine_parity.py -m gpu -x -q 2>&1 | tail -3
SWEEP="base:MH_X=0 rowoff:MH_CORR_WARP_ROW_MIN=0 row24:MH_CORR_WARP_ROW_MIN=24 row48:MH_CORR_WARP_ROW_MIN=48 row96:MH_CORR_WARP_ROW_MIN=96 base2:MH_X=0" bash scripts/gpu_sweep.sh $TAG
timeout 300 python scripts/plan_table.py > $OUT/plan_table_madnet.txt 2>&1; grep -E "corr_warp|kind 10|ops," $OUT/plan_table_madnet.txt | head
