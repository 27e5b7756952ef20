#!/bin/bash
# round 6, call r: staggered forward tiles, one box, three forms alternating four times: off (bit 6), 128-column layers only (bit 7), every 128-pixel one-per-CU tile
OUT=gpurun_out/r6r; mkdir -p $OUT
Q="--no-paths --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 50 --repeats 3"
for r in 1 2 3 4; do
for v in "off:--set tune.conv_planes=64" "on128:--set tune.conv_planes=128" "on:"; do
  n=${v%%:*}$r; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail r_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'])" || tail -5 $OUT/$n.err
done
done
timeout 300 python scripts/plan_table.py > $OUT/plan_table_on.txt 2>&1
timeout 300 python scripts/plan_table.py --tune conv_planes=64 > $OUT/plan_table_off.txt 2>&1
grep -n "conv_planes_kernel<" $OUT/plan_table_on.txt | sed -n 3,60p > $OUT/on_rows.txt
grep -n "conv_planes_kernel<" $OUT/plan_table_off.txt | sed -n 3,60p > $OUT/off_rows.txt
paste -d'\n' $OUT/off_rows.txt $OUT/on_rows.txt | grep -B1 staggered | cut -c1-130
