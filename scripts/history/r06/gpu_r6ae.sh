#!/bin/bash
# round 6, call ae: MADNet with the momentum update of a batch's layers behind the batch (EARLY_UPDATE, measured worse in round 4 #17) re-measured on today's step
OUT=gpurun_out/r6ae; mkdir -p $OUT
Q="--no-paths --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 50 --repeats 3 --no-cpu-baseline"
for r in 1 2 3; do
for v in "base:" "early:--set engine.EARLY_UPDATE=True"; do
  n=${v%%:*}$r; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail ae_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j['value'], j['config'].get('ops_per_step'))" || tail -3 $OUT/$n.err
done
done
