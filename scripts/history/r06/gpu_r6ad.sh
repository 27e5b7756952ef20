#!/bin/bash
# round 6, call ad: workgroups per streamed filter-gradient batch, finer, two alternations
OUT=gpurun_out/r6ad; mkdir -p $OUT
Q="--no-paths --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 50 --repeats 3 --no-cpu-baseline"
for r in 1 2; do
for v in "base:" "g128:--set ops.WGRAD_STREAM_WGS=128" "g160:--set ops.WGRAD_STREAM_WGS=160" "g192:--set ops.WGRAD_STREAM_WGS=192" "g224:--set ops.WGRAD_STREAM_WGS=224" "g192p130:--set ops.WGRAD_STREAM_WGS=192 --set tune.wgrad_target_pct=130"; do
  n=${v%%:*}$r; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail ad_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j['value'])" || tail -3 $OUT/$n.err
done
done
for v in "mad_base:--mode MAD" "mad_g192:--mode MAD --set ops.WGRAD_STREAM_WGS=192" "mad_base2:--mode MAD" "mad_g192b:--mode MAD --set ops.WGRAD_STREAM_WGS=192"; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail ad_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j['value'])" || tail -3 $OUT/$n.err
done
