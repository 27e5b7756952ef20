#!/bin/bash
TAG=${1:-r5i}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python scripts/exp/planes_phases_step.py 2>&1 | grep -v amdgpu.ids | tee $OUT/planes_phases_step.txt
