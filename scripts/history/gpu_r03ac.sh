#!/bin/bash
# round 2, call BC: new filter-gradient workgroup targets: parity + FULL / MAD / DispNet check
TAG=${1:-r03ac}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 5"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
timeout 900 python -m pytest tests/test_conv_parity.py tests/test_engine_parity.py tests/test_dispnet_parity.py -m gpu -x -q 2>&1 | tail -2
run full MH_X=1
run full150 MH_WGRAD_TARGET_PCT=150
EXTRA="--mode MAD" run mad MH_X=1
EXTRA="--mode MAD" run mad150 MH_WGRAD_TARGET_PCT=150
EXTRA="--model dispnet --steps 30" run dispnet MH_X=1
EXTRA="--model dispnet --steps 30" run dispnet150 MH_WGRAD_TARGET_PCT=150
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]])
    except Exception as ex: print(f, "ERR", ex)
PY
