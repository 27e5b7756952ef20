#!/bin/bash
# round 3, call Z: bf16-only gradient maps between shadow-staging input gradients + leaky masks from the activations' shadows (mh_conv2d_sh3)
TAG=${1:-r3z}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_parity.py -m gpu -x -q 2>&1 | tail -3
SWEEP="base:MH_X=0 noonly:MH_SHADOW_ONLY=0 base2:MH_X=0 noonly2:MH_SHADOW_ONLY=0" bash scripts/gpu_sweep.sh $TAG
timeout 300 python scripts/plan_table.py 2>&1 | grep -E "conv_patch_kernel<.*dgrad" | head -14
timeout 300 python bench.py --steps 50 --repeats 3 --no-paths --no-roofline --no-step-surface 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('epe', j.get('epe_vs_oracle'), j.get('within_tolerance'), j['ms_per_step'])"
