#!/bin/bash
# round 3, call T: head backward / forward fusion (mh_head_bwd, mh_conv2d_head): parity on the GPU, A/B in the step
TAG=${1:-r3t}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_parity.py -m gpu -x -q -k "head_bwd" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_engine_parity.py tests/test_api_gpu.py tests/test_dispnet_parity.py -m gpu -x -q 2>&1 | tail -3
SWEEP="base:MH_X=0 nofuse:MH_FUSE_HEAD=0 base2:MH_X=0 nofuse2:MH_FUSE_HEAD=0" bash scripts/gpu_sweep.sh $TAG
timeout 300 python scripts/plan_table.py > $OUT/plan_table_madnet.txt 2>&1; grep -E "head_bwd|ops,|kind 31|kind 30" $OUT/plan_table_madnet.txt | head
SWEEP="mad:MH_X=0 mad_nofuse:MH_FUSE_HEAD=0" BENCH_ARGS="--mode MAD" bash scripts/gpu_sweep.sh $TAG
