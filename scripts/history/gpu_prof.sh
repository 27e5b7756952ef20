#!/bin/bash
# usage: gpu_prof.sh TAG PRECISION  -- rocprofv3 kernel stats of the eager plan
TAG=${1:-r01x}; PREC=${2:-bf16}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o madnet -- python $GRAFT_REPO_ROOT/bench.py --precision $PREC --steps 10 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-parity-path --wgrad-lanes 0 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1)
tail -1 $OUT/prof.log | cut -c1-300
