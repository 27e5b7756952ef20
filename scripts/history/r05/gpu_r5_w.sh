#!/bin/bash
# round 5, call w: level_front_kernel<..,HEAD> with two fine rows per workgroup vs one (mh_tune_corr bit 3): GPU parity + kernel trace of the NONE step + same-box A/B
TAG=${1:-r5w}; OUT=gpurun_out/$TAG; mkdir -p $OUT; R=$(pwd); WORK=/tmp/prof_$TAG; mkdir -p $WORK; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_parity.py -q -m gpu -k "level_front or conv_image" 2>&1 | tail -1
P="--mode NONE --steps 10 --warmup 2 --repeats 1 --min-region-seconds 0 --no-cpu-baseline --no-roofline --no-paths --no-step-surface --no-configs --drift-steps 0"
for v in "rows2:" "rows1:--set tune.corr=9"; do
  n=${v%%:*}; f=${v#*:}
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $WORK/prof_$n -o $n -- python $R/bench.py $P $f > $R/$OUT/prof_$n.log 2>&1)
  k=$(find $WORK/prof_$n -name "*kernel_stats.csv" | head -1); [ -n "$k" ] && cp $k $OUT/none_${n}_kernel_stats.csv
  grep -i "level_front\|conv_image" $OUT/none_${n}_kernel_stats.csv | cut -c1-160
done
Q="--steps 200 --warmup 20 --no-configs --no-cpu-baseline --no-paths --drift-steps 0 --no-step-surface --no-roofline"
for i in 1 2 3; do
  for v in "rows2:" "rows1:--set tune.corr=9"; do
    n=${v%%:*}; f=${v#*:}
    timeout 300 python bench.py $Q $f 2>/dev/null | tail -1 > $OUT/bench_${n}_$i.json
    python -c "import json; d=json.loads(open('$OUT/bench_${n}_$i.json').read()); print('FULL $n #$i: %.4f ms/step' % d['ms_per_step'])"
  done
done
