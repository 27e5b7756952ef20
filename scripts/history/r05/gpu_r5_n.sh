#!/bin/bash
# round 5, call n: HEAD_IN_FRONT with the head's bank in registers: kernel trace of the NONE step (on / off) + FULL same-box A/B
TAG=${1:-r5n}; bash scripts/history/r05/gpu_r5_m.sh $TAG; OUT=gpurun_out/$TAG
timeout 300 python -m pytest tests/test_ops_parity.py -q -m gpu -k "level_front" 2>&1 | tail -2
for i in 1 2 3; do
  for v in on off; do
    if [ $v = on ]; then S=""; else S="--set engine.HEAD_IN_FRONT=0"; fi
    timeout 300 python bench.py --steps 200 --warmup 20 --no-configs --no-cpu-baseline --no-paths --drift-steps 0 --no-step-surface --no-roofline $S 2>/dev/null | tail -1 > $OUT/bench_${v}_$i.json
    python -c "import json; d=json.loads(open('$OUT/bench_${v}_$i.json').read()); print('FULL HEAD_IN_FRONT $v #$i: %.4f ms/step' % d['ms_per_step'])"
  done
done
