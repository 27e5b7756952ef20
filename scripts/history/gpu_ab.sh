#!/bin/bash
# In-situ A/B of the patch-kernel dispatch thresholds (whole-step bench, hipGraph replay).
TAG=${1:-ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --steps 150 --no-cpu-baseline --no-roofline --no-parity-path"
run() { name=$1; shift; env "$@" timeout 60 $B 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
run base MH_CONV_PATCH=1
run pix7680 MH_CONV_PATCH_MINPIX=7680
run cover160 MH_CONV_PATCH_COVER=160
run pix7680_cover160 MH_CONV_PATCH_MINPIX=7680 MH_CONV_PATCH_COVER=160
run pix1920 MH_CONV_PATCH_MINPIX=1920
run cover220 MH_CONV_PATCH_COVER=220
for f in $OUT/bench_*.json; do echo "$f: $(python -c 'import sys,json; d=json.load(open(sys.argv[1])); print(round(d["value"],1), round(d["ms_per_step"],4))' $f)"; done
