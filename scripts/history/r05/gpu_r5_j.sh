#!/bin/bash
TAG=${1:-r5j}; OUT=gpurun_out/$TAG; mkdir -p $OUT
bash scripts/gpu_pmc_r05.sh $TAG/pmc 2>&1 | tail -90
ls $OUT/pmc
