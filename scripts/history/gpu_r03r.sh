#!/bin/bash
# round 2, call AR: DispNet 'mixed' with the insensitive layers in plain bf16 (forward): tolerance test + A/B
TAG=${1:-r03r}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dispnet_parity.py -m gpu -x -q -s 2>&1 | grep -E "EPE vs oracle|passed|failed" 
for v in 1 0; do MH_DISPNET_MIXED_BF16=$v timeout 300 python bench.py --model dispnet --steps 30 --repeats 3 2>/dev/null | tail -1 > $OUT/bench_dispnet_mixed_bf16fwd$v.json; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e.get("epe_vs_oracle"), e.get("within_tolerance"))
PY
