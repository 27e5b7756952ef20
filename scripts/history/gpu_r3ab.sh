#!/bin/bash
# round 3, call AB: conv5 (32 -> 64, stride 2, 48x160 x 2 towers) forward on the small-layer bank kernel (split-bf16) instead of exact fp32 on the tiled kernel
TAG=${1:-r3ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
SWEEP="base:MH_X=0 s2_16k:MH_CONV_BANK_S2_MAXPIX=16384 base2:MH_X=0 s2_16k2:MH_CONV_BANK_S2_MAXPIX=16384" bash scripts/gpu_sweep.sh $TAG
MH_CONV_BANK_S2_MAXPIX=16384 timeout 300 python scripts/plan_table.py 2>&1 | grep -E "^ +[2-9] kind" | cut -c1-150
MH_CONV_BANK_S2_MAXPIX=16384 timeout 300 python bench.py --steps 50 --repeats 3 --no-paths --no-roofline --no-step-surface 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('epe', j.get('epe_vs_oracle'), j.get('within_tolerance'), j['ms_per_step'])"
