#!/bin/bash
# round 2, call I: critical-path trims (4-lane resize gradient, loss value / metrics and warp-gradient scatters on side lanes), x3 at 1/8 resolution,
# coarse estimators bf16 in the mixed forward pass.
TAG=${1:-r02i}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $OUT/pytest_gpu.txt
timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/bench_mixed.json
timeout 300 python bench.py $B --precision bf16 2>/dev/null | tail -1 > $OUT/bench_bf16.json
timeout 300 python bench.py $B --precision fp32 2>/dev/null | tail -1 > $OUT/bench_fp32.json
timeout 300 python bench.py $B --mode MAD 2>/dev/null | tail -1 > $OUT/bench_mad.json
timeout 600 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -1 > $OUT/bench_default_full.json
cat $OUT/pytest_gpu.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], e["timing"]["ms_per_step_all"], e["config"].get("ops_per_step"), e.get("epe_vs_oracle"))
PY
