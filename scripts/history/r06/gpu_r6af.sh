#!/bin/bash
# round 6, call af: the update of the estimators' / context network's parameters (73 %) at the cut where the pyramid's backward pass starts, on a lane of its own (Schedule.CUT_UPDATE)
OUT=gpurun_out/r6af; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_engine_parity.py tests/test_api_gpu.py tests/test_distributed_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
Q="--no-paths --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 50 --repeats 3 --no-cpu-baseline"
for r in 1 2 3 4; do
for v in "cut:" "late:--set engine.CUT_UPDATE=False"; do
  n=${v%%:*}$r; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail af_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j['value'], j.get('epe_vs_oracle'), j['config'].get('ops_per_step'))" || tail -3 $OUT/$n.err
done
done
for v in "b2b_cut:--step-sync none" "b2b_late:--step-sync none --set engine.CUT_UPDATE=False" "p4_cut:--concurrent-streams 4 --steps 30" "p4_late:--concurrent-streams 4 --steps 30 --set engine.CUT_UPDATE=False"; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail af_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j['value'])" || tail -3 $OUT/$n.err
done
