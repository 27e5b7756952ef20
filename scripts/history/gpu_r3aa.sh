#!/bin/bash
# round 3, call AA: patch-staged input gradient for the estimators' / context network's FIRST layers (38 / 33 gradient channels in rows of 40 / 36)
TAG=${1:-r3aa}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_parity.py -m gpu -x -q -k "from_the_shadow" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_engine_parity.py -m gpu -x -q 2>&1 | tail -3
SWEEP="base:MH_X=0 base2:MH_X=0" bash scripts/gpu_sweep.sh $TAG
timeout 300 python scripts/plan_table.py 2>&1 | grep -E "^ +(6[4-9]|7[0-9]|8[0-2]) kind" | cut -c1-150
timeout 300 python bench.py --steps 50 --repeats 3 --no-paths --no-roofline --no-step-surface 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('epe', j.get('epe_vs_oracle'), j.get('within_tolerance'), j['ms_per_step'])"
