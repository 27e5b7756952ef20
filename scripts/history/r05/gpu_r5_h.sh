#!/bin/bash
# round 5: same-box A/B of the small-layer bank kernel (previous prologue: scripts/exp/libmadnet_hip_prev.so, built from HEAD~1's conv_patch.hip) and of the
# row-owned correlation backward (tune.corr_row)
TAG=${1:-r5h}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
Q="--no-paths --no-cpu-baseline --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 20 --warmup 5 --repeats 3"
for i in 1 2 3; do
  for v in prev new; do
    if [ $v = prev ]; then export MADNET_HIP_LIB=$PWD/scripts/exp/libmadnet_hip_prev.so; else unset MADNET_HIP_LIB; fi
    timeout 300 python bench.py $Q 2>/dev/null | tail -1 > $OUT/bench_${v}_$i.json
    python -c "import json;d=json.load(open('$OUT/bench_${v}_$i.json'));print('$v', d['ms_per_step'], d['timing']['ms_per_step_all'])"
  done
done
unset MADNET_HIP_LIB
for v in 0 1 0 1; do
  timeout 300 python bench.py $Q --set tune.corr_row=$v 2>/dev/null | tail -1 > $OUT/bench_corr_row_$v.json
  python -c "import json;d=json.load(open('$OUT/bench_corr_row_$v.json'));print('corr_row=$v', d['ms_per_step'])"
done
for v in prev new; do
  if [ $v = prev ]; then export MADNET_HIP_LIB=$PWD/scripts/exp/libmadnet_hip_prev.so; else unset MADNET_HIP_LIB; fi
  timeout 300 python bench.py $Q --mode MAD 2>/dev/null | tail -1 > $OUT/bench_mad_$v.json; python -c "import json;d=json.load(open('$OUT/bench_mad_$v.json'));print('MAD $v', d['ms_per_step'])"
  timeout 300 python bench.py $Q --mode NONE 2>/dev/null | tail -1 > $OUT/bench_none_$v.json; python -c "import json;d=json.load(open('$OUT/bench_none_$v.json'));print('NONE $v', d['ms_per_step'])"
done
