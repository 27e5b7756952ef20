#!/bin/bash
# round 2, call AZ: lattice-cover limit of the patch / bank kernels (strongly dilated context layers)
TAG=${1:-r03y3}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
run cover125 MH_CONV_PATCH_COVER=125
run cover220 MH_CONV_PATCH_COVER=220
run cover260 MH_CONV_PATCH_COVER=260
run cover400 MH_CONV_PATCH_COVER=400
EXTRA="--precision bf16" run bf16_cover125 MH_CONV_PATCH_COVER=125
EXTRA="--precision bf16" run bf16_cover220 MH_CONV_PATCH_COVER=220
EXTRA="--mode MAD" run mad_cover220 MH_CONV_PATCH_COVER=220
EXTRA="--mode MAD" run mad_cover125 MH_CONV_PATCH_COVER=125
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]])
PY
