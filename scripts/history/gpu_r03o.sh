#!/bin/bash
# round 2, call AO: split-bf16 on the tiled implicit-GEMM kernel (forward layers without a patch / bank instance): parity + A/B (MADNet, DispNet)
TAG=${1:-r03o}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
timeout 900 python -m pytest tests/test_conv_parity.py tests/test_engine_parity.py tests/test_dispnet_parity.py -m gpu -x -q 2>&1 | tail -3
EXTRA="--no-paths" run x3_1 MH_CONV_X3_IGEMM=1
EXTRA="--no-paths" run x3_0 MH_CONV_X3_IGEMM=0
EXTRA="--no-paths" run x3_1b MH_CONV_X3_IGEMM=1
EXTRA="--model dispnet --steps 30" run dispnet_x3_1 MH_CONV_X3_IGEMM=1
EXTRA="--model dispnet --steps 30" run dispnet_x3_0 MH_CONV_X3_IGEMM=0
EXTRA="--mode MAD --no-paths" run mad_x3_1 MH_CONV_X3_IGEMM=1
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e["config"].get("ops_per_step"), e.get("epe_vs_oracle"))
PY
