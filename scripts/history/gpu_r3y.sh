#!/bin/bash
# round 3, call Y: 64x32 tile (4 waves) of the split-bf16 bank kernel for the <= 32-column layers with few workgroups
TAG=${1:-r3y}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
SWEEP="base:MH_X=0 t300:MH_CONV_BANK_TILE32_WGS=300 t100:MH_CONV_BANK_TILE32_WGS=100 base2:MH_X=0 t300b:MH_CONV_BANK_TILE32_WGS=300" bash scripts/gpu_sweep.sh $TAG
MH_CONV_BANK_TILE32_WGS=300 timeout 300 python scripts/plan_table.py 2>&1 | grep -E "conv_bank_kernel<(4,1,1,2|8,1,1,2)" | head
timeout 300 python scripts/plan_table.py 2>&1 | grep -E "conv_bank_kernel<(4,1,1,2|8,1,1,2)" | sed "s/^/off /" | head
MH_CONV_BANK_TILE32_WGS=300 timeout 300 python bench.py --steps 50 --repeats 3 --no-paths --no-roofline --no-step-surface 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('epe', j.get('epe_vs_oracle'), j.get('within_tolerance'), j['ms_per_step'])"
