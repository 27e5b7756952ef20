#!/bin/bash
# round 5, call v: conv_image_fwd_kernel variants timed alone (plan table line) + NONE step
TAG=${1:-r5v}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 300 python -m pytest tests/test_ops_parity.py -q -m gpu -k "conv_image" 2>&1 | tail -1
timeout 300 python scripts/plan_table.py 2>/dev/null | grep -i "conv_image\|pad_reflect" | head -4
Q="--steps 200 --warmup 20 --no-configs --no-cpu-baseline --no-paths --drift-steps 0 --no-step-surface --no-roofline"
for i in 1 2; do
  timeout 300 python bench.py $Q --mode NONE 2>/dev/null | tail -1 > $OUT/bench_none_$i.json
  python -c "import json; d=json.loads(open('$OUT/bench_none_$i.json').read()); print('NONE #$i: %.4f ms/step' % d['ms_per_step'])"
done
