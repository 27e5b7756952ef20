#!/bin/bash
# round 2, call AS: MADNet 'mixed': pyramid conv7..conv12 in plain bf16 (forward): tolerance + A/B
TAG=${1:-r03s}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
timeout 900 python -m pytest tests/test_engine_parity.py -m gpu -x -q 2>&1 | tail -2
run pyr7 MH_PYR_BF16_FROM=7
run pyr13 MH_PYR_BF16_FROM=13
run pyr9 MH_PYR_BF16_FROM=9
run pyr7b MH_PYR_BF16_FROM=7
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e.get("epe_vs_oracle"), e.get("within_tolerance"))
PY
