#!/bin/bash
# round 6, call d: DispNet with split-bf16 on the tiled kernel for conv1 / conv2 (mh_tune_conv_x3_igemm) against the exact-fp32 default
OUT=gpurun_out/r6d; mkdir -p $OUT
Q="--no-paths --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 50 --repeats 3 --model dispnet"
for v in "base:" "x3:--set tune.conv_x3_igemm=1" "base2:" "x3b:--set tune.conv_x3_igemm=1"; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail d_$n.json 2>/dev/null | tail -1 > $OUT/dispnet_$n.json
  python -c "import json;j=json.load(open('$OUT/dispnet_$n.json'));print('$n', j['ms_per_step'], j.get('epe_vs_oracle'), j['timing'])"
done
timeout 300 python scripts/plan_table.py --model dispnet > $OUT/plan_table_dispnet.txt 2>&1; head -30 $OUT/plan_table_dispnet.txt
