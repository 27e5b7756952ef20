#!/bin/bash
# round 2, call AX: stride-2 forward instances of the small-layer bank kernel (pyramid conv7 / 9 / 11): parity + A/B
TAG=${1:-r03x}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
timeout 900 python -m pytest tests/test_conv_parity.py tests/test_engine_parity.py -m gpu -x -q 2>&1 | tail -2
run s2_1 MH_CONV_BANK_SMALL_S2=1
run s2_0 MH_CONV_BANK_SMALL_S2=0
run s2_1b MH_CONV_BANK_SMALL_S2=1
run s2_0b MH_CONV_BANK_SMALL_S2=0
EXTRA="--mode MAD" run mad_s2_1 MH_CONV_BANK_SMALL_S2=1
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]])
PY
