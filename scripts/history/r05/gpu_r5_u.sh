#!/bin/bash
# round 5, call u: IMAGE_CONV with the padding launch on lane 1 in front of the loss reductions (no fork of its own): same-box A/B through the environment
# (MH_IMAGE_CONV reaches the engines built behind the Nets API too: MAD)
TAG=${1:-r5u}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_engine_parity.py tests/test_zz_wgrad_image.py -q -m gpu -k "full_step or mad_step or mixed or wgrad_image or image_layer or scheduling" 2>&1 | tail -2
Q="--steps 200 --warmup 20 --no-configs --no-cpu-baseline --no-paths --drift-steps 0 --no-step-surface --no-roofline"
for i in 1 2 3; do
  for mode in FULL MAD NONE; do
    for v in 1 0; do
      MH_IMAGE_CONV=$v timeout 300 python bench.py $Q --mode $mode 2>/dev/null | tail -1 > $OUT/bench_${mode}_${v}_$i.json
      python -c "import json; d=json.loads(open('$OUT/bench_${mode}_${v}_$i.json').read()); print('$mode MH_IMAGE_CONV=$v #$i: %.4f ms/step' % d['ms_per_step'])"
    done
  done
done
