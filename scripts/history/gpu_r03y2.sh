#!/bin/bash
# round 2, call AY: re-sweep of the patch-kernel pixel floor (bf16 input gradients at 1/8 resolution) and of the lattice-cover limit in the current step
TAG=${1:-r03y2}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
run base MH_X=1
run minpix7680 MH_CONV_PATCH_MINPIX=7680
run minpix15360 MH_CONV_PATCH_MINPIX=15360
run cover160 MH_CONV_PATCH_COVER=160
run cover220 MH_CONV_PATCH_COVER=220
run base2 MH_X=1
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]])
PY
