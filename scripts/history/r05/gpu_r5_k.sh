#!/bin/bash
# round 5, call k: the disparity heads inside the next level's front-end launch (Schedule.HEAD_IN_FRONT): parity on the GPU, then same-box A/B
TAG=${1:-r5k}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_ops_parity.py tests/test_engine_parity.py tests/test_ref_graph.py tests/test_abi.py -x -q -m gpu \
    -k "level_front or full_step or mad_step or mixed or ref_graph or abi or scheduling" > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
for i in 1 2 3; do
  for v in on off; do
    if [ $v = on ]; then S=""; else S="--set engine.HEAD_IN_FRONT=0"; fi
    timeout 300 python bench.py --steps 200 --warmup 20 --no-configs --no-cpu-baseline --no-paths --drift-steps 0 --no-step-surface $S > $OUT/bench_${v}_$i.json 2> $OUT/bench_${v}_$i.err
    python - <<PY
import json
d=json.loads(open("$OUT/bench_${v}_$i.json").read().strip().splitlines()[-1])
print("HEAD_IN_FRONT $v #$i: %.4f ms/step  epe %.3g  ops %s" % (d["ms_per_step"], d.get("epe_vs_oracle", float("nan")), d.get("plan_ops")))
PY
  done
done
for mode in none mad; do
  for v in on off; do
    if [ $v = on ]; then S=""; else S="--set engine.HEAD_IN_FRONT=0"; fi
    timeout 300 python bench.py --steps 200 --warmup 20 --no-configs --mode ${mode^^} --no-cpu-baseline --no-paths --drift-steps 0 $S > $OUT/bench_${mode}_$v.json 2> $OUT/bench_${mode}_$v.err
    python - <<PY
import json
d=json.loads(open("$OUT/bench_${mode}_$v.json").read().strip().splitlines()[-1])
print("$mode HEAD_IN_FRONT $v: %.4f ms/step" % d["ms_per_step"])
PY
  done
done
