#!/bin/bash
# round 2, call AA: first N filter-gradient batches launched undeferred (start early, one queue hop of the critical path each)
TAG=${1:-r03a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
run nd0 MH_NODEFER_BATCHES=0
run nd1 MH_NODEFER_BATCHES=1
run nd2 MH_NODEFER_BATCHES=2
run nd3 MH_NODEFER_BATCHES=3
run nd2_l2 MH_NODEFER_BATCHES=2 MH_WGRAD_LANES=2
run nd4_l2 MH_NODEFER_BATCHES=4 MH_WGRAD_LANES=2
run nd0_again MH_NODEFER_BATCHES=0
C="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 12 --warmup 3 --repeats 1"
(cd /tmp && MH_NODEFER_BATCHES=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_graph -o madnet -- python $GRAFT_REPO_ROOT/bench.py $C > $GRAFT_REPO_ROOT/$OUT/prof_graph.log 2>&1)
f=$(ls $OUT/prof_graph/*kernel_trace.csv | head -1)
python scripts/trace_timeline.py $f > $OUT/timeline_nd2.txt 2>&1
rm -rf $OUT/prof_graph
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e["config"].get("ops_per_step"))
PY
