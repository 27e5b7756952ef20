#!/bin/bash
TAG=${1:-r01b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 150 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/bench_graph.log 2>&1; echo "exit $?" >> $OUT/bench_graph.log
timeout 150 python bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline > $OUT/bench_roofline.log 2>&1; echo "exit $?" >> $OUT/bench_roofline.log
timeout 240 python bench.py --steps 5 --warmup 2 --no-graph --no-roofline > $OUT/bench_cpu.log 2>&1; echo "exit $?" >> $OUT/bench_cpu.log
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o madnet -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-graph --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1)
find $OUT/prof -type f | head -20 >> $OUT/prof.log
for f in $OUT/bench_graph.log $OUT/bench_roofline.log $OUT/bench_cpu.log; do echo "== $f"; tail -6 $f | cut -c1-1500; done
