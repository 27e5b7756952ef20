#!/bin/bash
# round 6, call n: the staggered form of the 128 x 128 plane tile (two out-of-phase wave groups, wave-local epilogue): parity on the GPU, whole-step A/B, per-op table
OUT=gpurun_out/r6n; mkdir -p $OUT
timeout 900 python -m pytest tests/test_conv_planes.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
Q="--no-paths --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 50 --repeats 3"
for v in "base:" "stg:--set tune.conv_planes=16" "stg_noprio:--set tune.conv_planes=16400" "base2:" "stg2:--set tune.conv_planes=16" "stg_noprio2:--set tune.conv_planes=16400"; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail n_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j.get('epe_vs_oracle'), j['config'].get('ops_per_step'))" || tail -5 $OUT/$n.err
done
timeout 300 python scripts/plan_table.py --tune conv_planes=16 > $OUT/plan_table_stg.txt 2>&1
timeout 300 python scripts/plan_table.py --tune conv_planes=16400 > $OUT/plan_table_stg_noprio.txt 2>&1
timeout 300 python scripts/plan_table.py > $OUT/plan_table_base.txt 2>&1
grep -n "conv_planes_kernel<" $OUT/plan_table_base.txt | sed -n 3,60p > $OUT/base_rows.txt
grep -n "conv_planes_kernel<" $OUT/plan_table_stg.txt | sed -n 3,60p > $OUT/stg_rows.txt
grep -n "conv_planes_kernel<" $OUT/plan_table_stg_noprio.txt | sed -n 3,60p > $OUT/stgnp_rows.txt
paste -d'\n' $OUT/base_rows.txt $OUT/stg_rows.txt | grep -B1 staggered | cut -c1-150
