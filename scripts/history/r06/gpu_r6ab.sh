#!/bin/bash
# round 6, call ab: the artifact run on the final commit (staggered tiles off by default) + one more same-box alternation of the forward-only staggered tile
bash scripts/gpu_final.sh r06final4
OUT=gpurun_out/r06final4
Q="--no-paths --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 50 --repeats 3 --no-cpu-baseline"
for r in 1 2 3; do
for v in "off:" "fwd_staggered:--set tune.conv_planes=16"; do
  n=${v%%:*}$r; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail ab_$n.json 2>>$OUT/ab.err | tail -1 > $OUT/ab_$n.json
  python -c "import json;j=json.load(open('$OUT/ab_$n.json'));print('$n', j['ms_per_step'], j['value'])"
done
done
