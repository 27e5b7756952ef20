#!/bin/bash
# round 3, call Z2: mask shadow as one 8-byte load; A/B of the three shadow options
TAG=${1:-r3z2}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
SWEEP="base:MH_X=0 noonly:MH_SHADOW_ONLY=0 nodgrad:MH_SHADOW_DGRAD=0 base2:MH_X=0 noonly2:MH_SHADOW_ONLY=0" bash scripts/gpu_sweep.sh $TAG
timeout 300 python scripts/plan_table.py 2>&1 | grep -E "^ +[0-9]+ kind.*conv_patch_kernel<.*dgrad" | head -12
