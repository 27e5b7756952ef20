#!/bin/bash
# round 6, call l: conv5's accumulating stride-2 input gradient (MADNet), conv3's input gradient on the 5x5 parity-class kernel (DispNet)
OUT=gpurun_out/r6l; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_conv_planes.py tests/test_dispnet_parity.py tests/test_ref_graph.py tests/test_engine_parity.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
Q="--no-paths --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 50 --repeats 3"
for v in "mad_acc:" "mad_noacc:--set engine.PLANES_S2_ACC=False" "mad_acc2:" "mad_noacc2:--set engine.PLANES_S2_ACC=False" "dn_on:--model dispnet" "dn_off:--model dispnet --set engine.PLANES_S2=False" "dn_on2:--model dispnet"; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail l_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j.get('epe_vs_oracle'), j['config'].get('ops_per_step'))" || tail -5 $OUT/$n.err
done
timeout 300 python scripts/plan_table.py --model dispnet > $OUT/plan_table_dispnet.txt 2>&1; head -14 $OUT/plan_table_dispnet.txt; grep -n "s2bwd" $OUT/plan_table_dispnet.txt | head
timeout 300 python scripts/plan_table.py > $OUT/plan_table_madnet.txt 2>&1; grep -n "s2bwd\|s2fwd" $OUT/plan_table_madnet.txt | head
timeout 300 python scripts/exp/det_probe.py 2>&1 | tail -2; timeout 300 python scripts/exp/det_probe.py --model dispnet 2>&1 | tail -2
