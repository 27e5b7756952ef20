#!/bin/bash
# round 3, call V: input gradients staging the bf16 shadow of dz (mh_conv2d_sh2), exact-2x head backward, stride-2 rows kernel with short blocks
TAG=${1:-r3v}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_parity.py tests/test_ops_parity.py -m gpu -x -q -k "from_the_shadow or head_bwd or conv_rows" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_engine_parity.py -m gpu -x -q 2>&1 | tail -3
SWEEP="base:MH_X=0 noshd:MH_SHADOW_DGRAD=0 base2:MH_X=0 noshd2:MH_SHADOW_DGRAD=0 rows32k:MH_CONV_ROWS_MINPIX=32768" bash scripts/gpu_sweep.sh $TAG
timeout 300 python scripts/plan_table.py > $OUT/plan_table_madnet.txt 2>&1; grep -E "ops,|kind 30|conv_patch_kernel<.*dgrad" $OUT/plan_table_madnet.txt | head -24
MH_SHADOW_DGRAD=0 timeout 300 python scripts/plan_table.py 2>&1 | grep -E "^ +[0-9]+ kind.*conv_patch_kernel<.*dgrad" | sed "s/^/noshd /"
MH_CONV_ROWS_MINPIX=32768 timeout 300 python scripts/plan_table.py 2>&1 | grep -E "^ +[0-9]+ kind.*conv_rows" | sed "s/^/32k /"
