#!/bin/bash
# round 5, call t: why is MAD slower with IMAGE_CONV?  kernel trace of the replayed MAD step, on / off
TAG=${1:-r5t}; OUT=gpurun_out/$TAG; mkdir -p $OUT; R=$(pwd); WORK=/tmp/prof_$TAG; mkdir -p $WORK; export TMPDIR=/tmp
P="--mode MAD --steps 10 --warmup 2 --repeats 1 --min-region-seconds 0 --no-cpu-baseline --no-roofline --no-paths --no-step-surface --no-configs --drift-steps 0"
for v in "on:" "off:--set engine.IMAGE_CONV=0"; do
  n=${v%%:*}; f=${v#*:}
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $WORK/prof_$n -o $n -- python $R/bench.py $P $f > $R/$OUT/prof_$n.log 2>&1)
  k=$(find $WORK/prof_$n -name "*kernel_stats.csv" | head -1); [ -n "$k" ] && cp $k $OUT/mad_${n}_kernel_stats.csv
  k=$(find $WORK/prof_$n -name "*kernel_trace.csv" | head -1); [ -n "$k" ] && python scripts/trace_timeline.py $k > $OUT/timeline_mad_$n.txt 2>&1
  head -1 $OUT/timeline_mad_$n.txt; tail -2 $OUT/timeline_mad_$n.txt
done
