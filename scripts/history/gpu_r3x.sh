#!/bin/bash
# round 3, call X: 64x64 tile (4 waves) of the split-bf16 bank kernel for under-filled grids (the 1/8-resolution level)
TAG=${1:-r3x}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_parity.py -m gpu -x -q -k "fragment_bank" 2>&1 | tail -3
SWEEP="base:MH_X=0 off:MH_CONV_BANK_SMALL_TILE_WGS=0 w130:MH_CONV_BANK_SMALL_TILE_WGS=130 base2:MH_X=0 off2:MH_CONV_BANK_SMALL_TILE_WGS=0" bash scripts/gpu_sweep.sh $TAG
timeout 300 python scripts/plan_table.py 2>&1 | grep -E "^ +(3[5-9]|4[0-1]) kind"
MH_CONV_BANK_SMALL_TILE_WGS=0 timeout 300 python scripts/plan_table.py 2>&1 | grep -E "^ +(3[5-9]|4[0-1]) kind" | sed "s/^/off /"
timeout 300 python bench.py --steps 50 --repeats 3 --no-paths --no-roofline --no-step-surface 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('epe', j.get('epe_vs_oracle'), j.get('within_tolerance'), j['ms_per_step'])"
