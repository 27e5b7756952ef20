#!/bin/bash
# round 6, call m: DispNet conv3 on the stride-2 plane kernels with the larger pragma-unroll budget -- input gradient (K16 = 16: 400 steps), forward two-row instance (K16 = 10: 250 steps)
OUT=gpurun_out/r6m; mkdir -p $OUT
timeout 900 python -m pytest tests/test_conv_planes.py tests/test_dispnet_parity.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
Q="--no-paths --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 50 --repeats 3"
for v in "dn_on:--model dispnet" "dn_off:--model dispnet --set engine.PLANES_S2=False" "dn_c3:--model dispnet --set engine.PLANES_S2_CONV3=True" "dn_on2:--model dispnet" "dn_c3b:--model dispnet --set engine.PLANES_S2_CONV3=True" "mad:" ; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail m_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j.get('epe_vs_oracle'), j['config'].get('ops_per_step'))" || tail -5 $OUT/$n.err
done
timeout 300 python scripts/plan_table.py --model dispnet > $OUT/plan_table_dispnet.txt 2>&1; head -12 $OUT/plan_table_dispnet.txt; grep -n "s2bwd\|s2fwd" $OUT/plan_table_dispnet.txt | head
