#!/bin/bash
# round 2, call AL: fused correlation + warp gradient per level (mh_corr_warp_bwd): parity + A/B
TAG=${1:-r03l}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
timeout 900 python -m pytest tests/test_ops_parity.py tests/test_engine_parity.py tests/test_api_gpu.py -m gpu -x -q 2>&1 | tail -3
run fuse1 MH_FUSE_BACK=1
run fuse0 MH_FUSE_BACK=0
run fuse1b MH_FUSE_BACK=1
run fuse0b MH_FUSE_BACK=0
EXTRA="--mode MAD" run mad_fuse1 MH_FUSE_BACK=1
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e["config"].get("ops_per_step"))
PY
