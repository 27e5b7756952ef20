#!/bin/bash
# round 6, call p: staggered tile on the forward layers only (the backward section is throughput-bound: r6o), and s_setprio 2 for the main lane's plane kernels
OUT=gpurun_out/r6p; mkdir -p $OUT
Q="--no-paths --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 50 --repeats 3"
for r in 1 2; do
for v in "base:" "f16:--set tune.conv_planes=16" "prio:--set tune.conv_planes=32768" "f16prio:--set tune.conv_planes=32784" "both:--set tune.conv_planes=48"; do
  n=${v%%:*}$r; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail p_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j.get('epe_vs_oracle'), j['config'].get('ops_per_step'))" || tail -5 $OUT/$n.err
done
done
for v in "none_base:--mode NONE" "none_f16:--mode NONE --set tune.conv_planes=16" "mad_base:--mode MAD" "mad_f16:--mode MAD --set tune.conv_planes=16" "mad_both:--mode MAD --set tune.conv_planes=48" "none_base2:--mode NONE" "none_f16b:--mode NONE --set tune.conv_planes=16"; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail p_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j.get('epe_vs_oracle'), j['config'].get('ops_per_step'))" || tail -5 $OUT/$n.err
done
