#!/bin/bash
# round 6, call ai: how many of the first filter-gradient batches are launched at once (NODEFER_BATCHES) re-swept after the executor change
OUT=gpurun_out/r6ai; mkdir -p $OUT
Q="--no-paths --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 50 --repeats 3 --no-cpu-baseline"
for r in 1 2; do
for v in "base:" "n0:--set engine.NODEFER_BATCHES=0" "n1:--set engine.NODEFER_BATCHES=1" "n3:--set engine.NODEFER_BATCHES=3" "n9:--set engine.NODEFER_BATCHES=9"; do
  n=${v%%:*}$r; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail ai_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j['value'])" || tail -3 $OUT/$n.err
done
done
