#!/bin/bash
# round 3, call AD: the row kernel's input gradient (conv2) with dz and the mask from bf16 shadows; pyramid input gradients pass their shadows
TAG=${1:-r3ad}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_parity.py -m gpu -x -q -k "conv_rows" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_engine_parity.py -m gpu -x -q 2>&1 | tail -2
SWEEP="base:MH_X=0 base2:MH_X=0" bash scripts/gpu_sweep.sh $TAG
timeout 300 python scripts/plan_table.py 2>&1 | grep -E "^ +1(3[0-9]|4[0-4]) kind" | cut -c1-150
timeout 300 python bench.py --steps 50 --repeats 3 --no-paths --no-roofline --no-step-surface 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('epe', j.get('epe_vs_oracle'), j.get('within_tolerance'), j['ms_per_step'])"
