#!/bin/bash
# round 6, call w: DispNet -- the filter-gradient side lane is the critical path of the backward pass (timeline: 1164 us back to back on ONE lane, the main lane waits 662 us
# at the end): lanes x batch size re-swept; + the producer-written shadows (PRODUCER_SHADOWS) A/B and parity
OUT=gpurun_out/r6w; mkdir -p $OUT
timeout 900 python -m pytest tests/test_dispnet_parity.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
Q="--model dispnet --no-paths --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 30 --repeats 3 --no-cpu-baseline"
for v in "base:" "noprod:--set engine.PRODUCER_SHADOWS=False" "l2:--set engine.SIDE_LANES=2" "l3:--set engine.SIDE_LANES=3" "l4:--set engine.SIDE_LANES=4" \
         "l2f1:--set engine.SIDE_LANES=2 --set engine.FLUSH_MIN=1" "l3f1:--set engine.SIDE_LANES=3 --set engine.FLUSH_MIN=1" "l4f1:--set engine.SIDE_LANES=4 --set engine.FLUSH_MIN=1" \
         "l2f3:--set engine.SIDE_LANES=2 --set engine.FLUSH_MIN=3" "l3f3:--set engine.SIDE_LANES=3 --set engine.FLUSH_MIN=3" "l1f1:--set engine.FLUSH_MIN=1" "base2:"; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail w_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j['value'], j.get('epe_vs_oracle'), j['config'].get('ops_per_step'))" || tail -5 $OUT/$n.err
done
