#!/bin/bash
# round 2, call O: grouped filter gradients for the small layers only + earlier issue of the last pyramid filter gradients
TAG=${1:-r02o}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
for i in 1 2 3; do timeout 300 python -m pytest tests/test_engine_parity.py -m gpu -x -q -k two_host 2>&1 | tail -15 > $OUT/pytest_two_threads_$i.txt; tail -1 $OUT/pytest_two_threads_$i.txt; done
run g1_tail32 MH_WGRAD_GROUP=1
run g0_tail32 MH_WGRAD_GROUP=0
run g1_tailnone MH_WGRAD_GROUP=1 MH_PYR_TAIL_FLUSH=
run g0_tailnone MH_WGRAD_GROUP=0 MH_PYR_TAIL_FLUSH=
run g1_tail2 MH_WGRAD_GROUP=1 MH_PYR_TAIL_FLUSH=2
run g1_m8k_tail32 MH_WGRAD_GROUP=1 MH_WGRAD_GROUP_MAXM=8192
run g1_m2k_tail32 MH_WGRAD_GROUP=1 MH_WGRAD_GROUP_MAXM=2048
run g1_tail32_again MH_WGRAD_GROUP=1
C="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 12 --warmup 3 --repeats 1"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_graph -o madnet -- python $GRAFT_REPO_ROOT/bench.py $C > $GRAFT_REPO_ROOT/$OUT/prof_graph.log 2>&1)
f=$(ls $OUT/prof_graph/*kernel_trace.csv | head -1)
python scripts/trace_timeline.py $f > $OUT/timeline.txt 2>&1
rm -rf $OUT/prof_graph
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e["config"].get("ops_per_step"))
PY
