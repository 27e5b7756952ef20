#!/bin/bash
# round 6, call q: forward layers on the staggered tile by default (+ the 96- and 64-column instances): A/B against bit 6 (off), GPU parity
OUT=gpurun_out/r6q; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_conv_planes.py tests/test_engine_parity.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
Q="--no-paths --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 50 --repeats 3"
for r in 1 2 3; do
for v in "on:" "off:--set tune.conv_planes=64"; do
  n=${v%%:*}$r; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail q_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j.get('epe_vs_oracle'), j['config'].get('ops_per_step'))" || tail -5 $OUT/$n.err
done
done
for v in "dn_on:--model dispnet" "dn_off:--model dispnet --set tune.conv_planes=64" "dn_on2:--model dispnet" "dn_off2:--model dispnet --set tune.conv_planes=64"; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail q_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j.get('epe_vs_oracle'), j['config'].get('ops_per_step'))" || tail -5 $OUT/$n.err
done
timeout 300 python scripts/plan_table.py > $OUT/plan_table.txt 2>&1; grep -n "staggered" $OUT/plan_table.txt | cut -c1-160
