#!/bin/bash
# In-situ A/B of the patch-kernel dispatch (whole-step bench, hipGraph replay): both directions / forward only / no generic-K dgrad.
TAG=${1:-ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --steps 150 --no-cpu-baseline --no-roofline --no-parity-path"
for m in 1 4097 8193 0; do MH_CONV_PATCH=$m timeout 60 $B 2>/dev/null | tail -1 > $OUT/bench_m$m.json; done
MH_CONV_PATCH=1 timeout 60 $B --wgrad-lanes 1 2>/dev/null | tail -1 > $OUT/bench_m1_lanes1.json
MH_CONV_PATCH=8193 timeout 60 $B 2>/dev/null | tail -1 > $OUT/bench_m8193_b.json
MH_CONV_PATCH=1 timeout 60 $B 2>/dev/null | tail -1 > $OUT/bench_m1_b.json
for f in $OUT/bench_*.json; do echo "$f: $(python -c 'import sys,json; d=json.load(open(sys.argv[1])); print(round(d["value"],1), round(d["ms_per_step"],4))' $f)"; done
