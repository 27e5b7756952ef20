#!/bin/bash
# round 5, call m: kernel trace of the replayed NONE step with and without HEAD_IN_FRONT (where do the four removed launches go?)
TAG=${1:-r5m}; OUT=gpurun_out/$TAG; mkdir -p $OUT; R=$(pwd); WORK=/tmp/prof_$TAG; mkdir -p $WORK; export TMPDIR=/tmp
P="--mode NONE --steps 10 --warmup 2 --repeats 1 --min-region-seconds 0 --no-cpu-baseline --no-roofline --no-paths --no-step-surface --no-configs --drift-steps 0"
for v in "on:" "off:--set engine.HEAD_IN_FRONT=0"; do
  n=${v%%:*}; f=${v#*:}
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $WORK/prof_$n -o $n -- python $R/bench.py $P $f > $R/$OUT/prof_$n.log 2>&1)
  k=$(find $WORK/prof_$n -name "*kernel_stats.csv" | head -1); [ -n "$k" ] && cp $k $OUT/none_${n}_kernel_stats.csv
  k=$(find $WORK/prof_$n -name "*kernel_trace.csv" | head -1); [ -n "$k" ] && python scripts/trace_timeline.py $k > $OUT/timeline_none_$n.txt 2>&1
  grep -i "level_front\|conv_n1" $OUT/none_${n}_kernel_stats.csv | cut -c1-200
  head -1 $OUT/timeline_none_$n.txt
done
