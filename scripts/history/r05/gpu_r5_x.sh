#!/bin/bash
# round 5, call x: conv3 (16 -> 32, stride 2, 61440 output pixels) on the row-streaming kernel (split-bf16) instead of the exact-fp32 tiled kernel: mh_tune_conv_rows(60000)
TAG=${1:-r5x}; OUT=gpurun_out/$TAG; mkdir -p $OUT
Q="--steps 200 --warmup 20 --no-configs --no-cpu-baseline --no-paths --drift-steps 0 --no-step-surface --no-roofline"
for i in 1 2 3; do
  for v in "base:" "rows:--set tune.conv_rows=60000"; do
    n=${v%%:*}; f=${v#*:}
    timeout 300 python bench.py $Q $f 2>/dev/null | tail -1 > $OUT/bench_${n}_$i.json
    python -c "import json; d=json.loads(open('$OUT/bench_${n}_$i.json').read()); print('FULL $n #$i: %.4f ms/step epe %.3g' % (d['ms_per_step'], d['epe_vs_oracle']))"
  done
done
timeout 300 python scripts/plan_table.py 2>/dev/null | grep -n "^ *[0-9]* kind" | sed -n 1,8p
