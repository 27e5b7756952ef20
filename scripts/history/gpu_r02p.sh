#!/bin/bash
# round 2, call P: fragment-bank split-bf16 forward kernel: parity, microbench, in-situ A/B, timeline
TAG=${1:-r02p}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
timeout 900 python -m pytest tests/test_conv_parity.py tests/test_engine_parity.py -m gpu -x -q 2>&1 | tail -5 > $OUT/pytest_gpu.txt
cat $OUT/pytest_gpu.txt
timeout 300 python scripts/microbench.py bank > $OUT/microbench_bank.txt 2>&1; cat $OUT/microbench_bank.txt
run bank1 MH_CONV_BANK=1
run bank0 MH_CONV_BANK=0
run bank1_again MH_CONV_BANK=1
C="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 12 --warmup 3 --repeats 1"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_graph -o madnet -- python $GRAFT_REPO_ROOT/bench.py $C > $GRAFT_REPO_ROOT/$OUT/prof_graph.log 2>&1)
f=$(ls $OUT/prof_graph/*kernel_trace.csv | head -1)
python scripts/trace_timeline.py $f > $OUT/timeline.txt 2>&1
rm -rf $OUT/prof_graph
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e["config"].get("ops_per_step"), e.get("epe_vs_oracle"))
PY
