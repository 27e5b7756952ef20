#!/bin/bash
# round 6, call ac: waves per workgroup and workgroups per batch of the streamed filter gradient re-swept (tuned in round 3, before the plane kernels and the first-writer back end)
OUT=gpurun_out/r6ac; mkdir -p $OUT
Q="--no-paths --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 50 --repeats 3 --no-cpu-baseline"
for v in "base:" "w6:--set ops.WGRAD_STREAM_WAVES=6" "w8:--set ops.WGRAD_STREAM_WAVES=8" "g192:--set ops.WGRAD_STREAM_WGS=192" "g320:--set ops.WGRAD_STREAM_WGS=320" "g384:--set ops.WGRAD_STREAM_WGS=384" \
         "w8g128:--set ops.WGRAD_STREAM_WAVES=8 --set ops.WGRAD_STREAM_WGS=128" "w6g192:--set ops.WGRAD_STREAM_WAVES=6 --set ops.WGRAD_STREAM_WGS=192" "base2:" "pct80:--set tune.wgrad_target_pct=80" "pct130:--set tune.wgrad_target_pct=130"; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail ac_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j['value'])" || tail -3 $OUT/$n.err
done
