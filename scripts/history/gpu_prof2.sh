#!/bin/bash
# rocprofv3 evidence for the default bench (kernel stats under hipGraph replay) + HBM-traffic PMC passes of the roofline kernels.
TAG=${1:-prof2}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_default -o madnet -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity-path > $GRAFT_REPO_ROOT/$OUT/prof_default.log 2>&1)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 60 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$c -o pmc -- python $GRAFT_REPO_ROOT/scripts/pmc_kernels.py > $GRAFT_REPO_ROOT/$OUT/$c.log 2>&1
done
cd $GRAFT_REPO_ROOT; find $OUT -name "*.csv" | head; tail -1 $OUT/prof_default.log | cut -c1-300
