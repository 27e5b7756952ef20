#!/bin/bash
TAG=${1:-r02h}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
timeout 600 python -m pytest tests/test_ops_parity.py tests/test_conv_parity.py tests/test_engine_parity.py -m gpu -x -q 2>&1 | tail -3 > $OUT/pytest_gpu.txt
timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/bench_mixed.json
timeout 300 python bench.py $B --precision bf16 2>/dev/null | tail -1 > $OUT/bench_bf16.json
timeout 300 python bench.py $B --precision fp32 2>/dev/null | tail -1 > $OUT/bench_fp32.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_serial -o madnet -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --repeats 1 --no-graph --no-cpu-baseline --no-roofline --no-paths --no-step-surface --wgrad-lanes 0 > $GRAFT_REPO_ROOT/$OUT/prof_serial.log 2>&1)
cat $OUT/pytest_gpu.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], e["timing"]["ms_per_step_all"], e["config"]["ops_per_step"])
PY
