#!/bin/bash
# graph-replay kernel traces of two variants of the step (rocprofv3 --kernel-trace; per-kernel durations INSIDE the replayed graph):
#   usage: gpurun -- 'bash scripts/gpu_trace_ab.sh tag "ENV_A=.. " "ENV_B=.."'   -> gpurun_out/<tag>/timeline_{a,b}.txt
TAG=${1:-trace}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; R=$PWD
export TMPDIR=/tmp
for v in a b; do
  if [ $v = a ]; then E="$2"; else E="$3"; fi
  W=/tmp/prof_$v; rm -rf $W
  (cd /tmp && env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $W -o madnet -- python $R/bench.py --steps 10 --warmup 2 --repeats 1 --min-region-seconds 0 --no-cpu-baseline --no-roofline --no-paths --no-step-surface $BENCH_ARGS > $OUT/prof_$v.log 2>&1)
  f=$(find $W -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/trace_timeline.py $f > $OUT/timeline_$v.txt 2>&1
  f=$(find $W -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_$v.csv
done
tail -n 3 $OUT/timeline_a.txt; tail -n 3 $OUT/timeline_b.txt
