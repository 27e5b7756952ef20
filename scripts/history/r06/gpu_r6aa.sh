#!/bin/bash
# round 6, call aa: kernel traces of the MADNet FULL replay queued back to back and with the host wait per step: where do the 21 us go?
OUT=gpurun_out/r6aa; mkdir -p $OUT; R=$(pwd); WORK=/tmp/r6aa; mkdir -p $WORK
P="--steps 30 --warmup 5 --repeats 1 --min-region-seconds 0 --no-cpu-baseline --no-roofline --no-paths --no-step-surface --no-configs --drift-steps 0"
for v in "b2b:--step-sync none" "sync:--step-sync stream"; do
  n=${v%%:*}; f=${v#*:}
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $WORK/prof_$n -o $n -- python $R/bench.py $P $f > $R/$OUT/prof_$n.log 2>&1)
  tail -1 $OUT/prof_$n.log | cut -c1-200
done
A=$(find $WORK/prof_b2b -name "*kernel_trace.csv" | head -1); B=$(find $WORK/prof_sync -name "*kernel_trace.csv" | head -1)
python scripts/exp/trace_compare.py $A $B | tee $OUT/trace_compare.txt
python scripts/trace_timeline.py $A pack_weights > $OUT/timeline_b2b.txt 2>&1; python scripts/trace_timeline.py $B pack_weights > $OUT/timeline_sync.txt 2>&1
tail -2 $OUT/timeline_b2b.txt; tail -2 $OUT/timeline_sync.txt
