#!/bin/bash
# round 3, call AC: the early filter-gradient batches (context, estimator 2, 3) on fewer workgroups (less split workspace, fewer CUs taken from the main chain)
TAG=${1:-r3ac}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
SWEEP="base:MH_X=0 e128:MH_WGRAD_EARLY_WGS=128 e192:MH_WGRAD_EARLY_WGS=192 e128b2:MH_WGRAD_EARLY_WGS=128,MH_WGRAD_EARLY_BATCHES=2 e96:MH_WGRAD_EARLY_WGS=96 base2:MH_X=0" bash scripts/gpu_sweep.sh $TAG
