#!/bin/bash
# round 2, call M: filter gradients accumulated with fp32 atomics straight into g (no workspace, no reduction launches) vs split workspace
TAG=${1:-r02m}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
run ws_l2 MH_WGRAD_ATOMIC=0
run atomic_l2 MH_WGRAD_ATOMIC=1
EXTRA="--wgrad-lanes 3" run atomic_l3 MH_WGRAD_ATOMIC=1
EXTRA="--wgrad-lanes 1" run atomic_l1 MH_WGRAD_ATOMIC=1
EXTRA="--wgrad-lanes 0" run atomic_l0 MH_WGRAD_ATOMIC=1
EXTRA="--wgrad-lanes 0" run ws_l0 MH_WGRAD_ATOMIC=0
EXTRA="--mode MAD" run mad_atomic MH_WGRAD_ATOMIC=1
EXTRA="--mode MAD" run mad_ws MH_WGRAD_ATOMIC=0
run ws_l2_again MH_WGRAD_ATOMIC=0
C="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 12 --warmup 3 --repeats 1"
(cd /tmp && MH_WGRAD_ATOMIC=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_graph -o madnet -- python $GRAFT_REPO_ROOT/bench.py $C > $GRAFT_REPO_ROOT/$OUT/prof_graph.log 2>&1)
f=$(ls $OUT/prof_graph/*kernel_trace.csv | head -1)
python scripts/trace_timeline.py $f > $OUT/timeline_atomic.txt 2>&1
rm -rf $OUT/prof_graph
MH_WGRAD_ATOMIC=1 timeout 600 python -m pytest tests/test_engine_parity.py -m gpu -x -q 2>&1 | tail -3 > $OUT/pytest_atomic.txt
cat $OUT/pytest_atomic.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e["config"].get("ops_per_step"))
PY
