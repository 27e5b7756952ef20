#!/bin/bash
# round 2, call AJ: one zero fill for all level feature gradients + the gradient-buffer fill on the filter-gradient lane: parity + A/B
TAG=${1:-r03j}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
timeout 900 python -m pytest tests/test_engine_parity.py tests/test_api_gpu.py tests/test_cli_gpu.py -m gpu -x -q 2>&1 | tail -3
run onefill1 MH_ONE_FILL=1
run onefill0 MH_ONE_FILL=0
run onefill1b MH_ONE_FILL=1
run onefill0b MH_ONE_FILL=0
EXTRA="--mode MAD" run mad_onefill1 MH_ONE_FILL=1
EXTRA="--mode MAD" run mad_onefill0 MH_ONE_FILL=0
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e["config"].get("ops_per_step"))
PY
