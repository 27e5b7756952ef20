#!/bin/bash
# round 5, call s: conv1 straight from the frames (mh_conv_image_fwd, Schedule.IMAGE_CONV): GPU parity subset, then same-box A/B (FULL x3, NONE, MAD)
TAG=${1:-r5s}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_parity.py tests/test_engine_parity.py tests/test_ref_graph.py tests/test_abi.py tests/test_api_gpu.py tests/test_cli_gpu.py -q -m gpu \
    -k "conv_image or level_front or full_step or mad_step or mixed or ref_graph or abi or scheduling or adapter or factory or cli or deterministic or private" > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt
Q="--steps 200 --warmup 20 --no-configs --no-cpu-baseline --no-paths --drift-steps 0 --no-step-surface --no-roofline"
for i in 1 2 3; do
  for v in on off; do
    if [ $v = on ]; then S=""; else S="--set engine.IMAGE_CONV=0"; fi
    timeout 300 python bench.py $Q $S 2>/dev/null | tail -1 > $OUT/bench_${v}_$i.json
    python -c "import json; d=json.loads(open('$OUT/bench_${v}_$i.json').read()); print('FULL IMAGE_CONV $v #$i: %.4f ms/step  epe %s' % (d['ms_per_step'], d.get('epe_vs_oracle')))"
  done
done
for mode in NONE MAD; do
  for i in 1 2; do
  for v in on off; do
    if [ $v = on ]; then S=""; else S="--set engine.IMAGE_CONV=0"; fi
    timeout 300 python bench.py $Q --mode $mode $S 2>/dev/null | tail -1 > $OUT/bench_${mode}_${v}_$i.json
    python -c "import json; d=json.loads(open('$OUT/bench_${mode}_${v}_$i.json').read()); print('$mode IMAGE_CONV $v #$i: %.4f ms/step' % d['ms_per_step'])"
  done
  done
done
timeout 300 python scripts/plan_table.py 2>/dev/null | grep -i "conv_image\|pad_reflect\|conv_rows" | head
