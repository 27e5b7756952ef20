#!/bin/bash
# round 5, call p: how many fork edges the estimators' filter-gradient batches are worth (Schedule.EST_FLUSH_AFTER), same box, alternating
TAG=${1:-r5p}; OUT=gpurun_out/$TAG; mkdir -p $OUT
Q="--steps 200 --warmup 20 --no-configs --no-cpu-baseline --no-paths --drift-steps 0 --no-step-surface --no-roofline"
for i in 1 2 3; do
  for v in "base:" "f236:--set engine.EST_FLUSH_AFTER=(2,3,6)" "f26:--set engine.EST_FLUSH_AFTER=(2,6)" "f2346:--set engine.EST_FLUSH_AFTER=(2,3,4,6)" "f6:--set engine.EST_FLUSH_AFTER=(6,)"; do
    n=${v%%:*}; f=${v#*:}
    timeout 300 python bench.py $Q $f 2>/dev/null | tail -1 > $OUT/bench_${n}_$i.json
    python -c "import json; d=json.loads(open('$OUT/bench_${n}_$i.json').read()); print('$n #$i: %.4f ms/step' % d['ms_per_step'])"
  done
done
