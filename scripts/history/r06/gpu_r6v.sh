#!/bin/bash
# round 6, call v: why are back-to-back replays 21 us slower than replay + host wait?  Two / three executable graphs of the same plan launched in turn
OUT=gpurun_out/r6v; mkdir -p $OUT
Q="--no-paths --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 50 --repeats 3 --no-cpu-baseline"
for r in 1 2; do
for v in "b2b:" "sync:--step-sync stream" "b2b_2g:--graph-copies 2" "b2b_3g:--graph-copies 3" "sync_2g:--step-sync stream --graph-copies 2"; do
  n=${v%%:*}$r; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail v_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j['value'])" || tail -5 $OUT/$n.err
done
done
