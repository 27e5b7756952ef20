#!/bin/bash
# round 6, call o: why the staggered plane tile is faster stand-alone (-15 us over ten ops) and slower inside the step (+15 us): kernel traces of the replayed graph, both forms
OUT=gpurun_out/r6o; mkdir -p $OUT; R=$(pwd); WORK=/tmp/r6o; mkdir -p $WORK
P="--steps 10 --warmup 2 --repeats 1 --min-region-seconds 0 --no-cpu-baseline --no-roofline --no-paths --no-step-surface --no-configs --drift-steps 0"
for v in "base:" "stg:--set tune.conv_planes=16"; do
  n=${v%%:*}; f=${v#*:}
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $WORK/prof_$n -o $n -- python $R/bench.py $P $f > $R/$OUT/prof_$n.log 2>&1)
  k=$(find $WORK/prof_$n -name "*kernel_trace.csv" | head -1); [ -n "$k" ] && python scripts/trace_timeline.py $k > $OUT/graph_timeline_$n.txt 2>&1
  head -1 $OUT/graph_timeline_$n.txt
done
