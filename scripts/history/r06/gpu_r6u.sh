#!/bin/bash
# round 6, call u: K steps enqueued back to back against a host wait after every step (bench.py --step-sync), FULL / MAD-free modes, alternating
OUT=gpurun_out/r6u; mkdir -p $OUT
Q="--no-paths --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 50 --repeats 3 --no-cpu-baseline"
for r in 1 2 3; do
for v in "b2b:" "sync:--step-sync stream"; do
  n=${v%%:*}$r; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail u_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j['value'])" || tail -5 $OUT/$n.err
done
done
for v in "dn_b2b:--model dispnet" "dn_sync:--model dispnet --step-sync stream" "none_b2b:--mode NONE" "none_sync:--mode NONE --step-sync stream" "p4_b2b:--concurrent-streams 4 --steps 30" "p4_sync:--concurrent-streams 4 --steps 30 --step-sync stream" "b4_b2b:--streams-per-gpu 4 --steps 30" "b4_sync:--streams-per-gpu 4 --steps 30 --step-sync stream"; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail u_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j['value'])" || tail -5 $OUT/$n.err
done
