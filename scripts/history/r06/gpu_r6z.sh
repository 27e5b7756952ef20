#!/bin/bash
# round 6, call z: streamed filter gradient with the three taps of a row sharing one run of transposing LDS reads (22 / 28 instead of 40 reads per step): parity, per-op table, step A/B
OUT=gpurun_out/r6z; mkdir -p $OUT
timeout 900 python -m pytest tests/test_wgrad_stream.py tests/test_engine_parity.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
for t in 0 16; do
  timeout 300 python scripts/plan_table.py --tune wgrad_stream=$t > $OUT/plan_table_$t.txt 2>&1
  echo "tune $t"; grep -n "kind 29" $OUT/plan_table_$t.txt | cut -c14-60 | tr '\n' ';'; echo
done
Q="--no-paths --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 50 --repeats 3 --no-cpu-baseline"
for r in 1 2 3; do
for v in "new:" "old:--set tune.wgrad_stream=16"; do
  n=${v%%:*}$r; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail z_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j['value'], j.get('epe_vs_oracle'))" || tail -5 $OUT/$n.err
done
done
for v in "dn_new:--model dispnet" "dn_old:--model dispnet --set tune.wgrad_stream=16" "mad_new:--mode MAD" "mad_old:--mode MAD --set tune.wgrad_stream=16"; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail z_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j['value'])" || tail -5 $OUT/$n.err
done
