#!/bin/bash
# round 2, call K: LDS-free small-layer kernel (conv_direct.hip): parity, in-situ A/B, serial per-kernel trace
TAG=${1:-r02k}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
timeout 900 python -m pytest tests/test_conv_parity.py tests/test_engine_parity.py tests/test_api_gpu.py -m gpu -x -q 2>&1 | tail -5 > $OUT/pytest_gpu.txt
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
run mixed_direct1 MH_X=1
run mixed_direct0 MH_CONV_DIRECT=0

timeout 300 python bench.py $B --precision bf16 2>/dev/null | tail -1 > $OUT/bench_bf16_direct1.json
MH_CONV_DIRECT=0 timeout 300 python bench.py $B --precision bf16 2>/dev/null | tail -1 > $OUT/bench_bf16_direct0.json
timeout 300 python bench.py $B --mode MAD 2>/dev/null | tail -1 > $OUT/bench_mad.json
run mixed_direct1_again MH_X=1

(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_serial -o madnet -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --repeats 1 --no-graph --no-cpu-baseline --no-roofline --no-paths --no-step-surface --wgrad-lanes 0 > $GRAFT_REPO_ROOT/$OUT/prof_serial.log 2>&1)
cat $OUT/pytest_gpu.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e["config"].get("ops_per_step"))
PY
grep -E "conv_direct|transpose_weights" $OUT/prof_serial/madnet_kernel_stats.csv | cut -c1-200
