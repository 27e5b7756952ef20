#!/bin/bash
# round 5, call l: HEAD_IN_FRONT -- the GPU parity subset in full, then NONE / MAD same-box A/B (three alternations)
TAG=${1:-r5l}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_parity.py tests/test_engine_parity.py tests/test_ref_graph.py tests/test_abi.py tests/test_api_gpu.py -q -m gpu \
    -k "level_front or full_step or mad_step or mixed or ref_graph or abi or scheduling or adapter or factory" > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt
for i in 1 2 3; do
for mode in NONE MAD; do
  for v in on off; do
    if [ $v = on ]; then S=""; else S="--set engine.HEAD_IN_FRONT=0"; fi
    timeout 300 python bench.py --steps 300 --warmup 20 --no-configs --mode $mode --no-cpu-baseline --no-paths --drift-steps 0 --no-step-surface --no-roofline $S > $OUT/bench_${mode}_${v}_$i.json 2> $OUT/bench_${mode}_${v}_$i.err
    python - <<PY
import json
d=json.loads(open("$OUT/bench_${mode}_${v}_$i.json").read().strip().splitlines()[-1])
print("$mode HEAD_IN_FRONT $v #$i: %.4f ms/step" % d["ms_per_step"])
PY
  done
done
done
