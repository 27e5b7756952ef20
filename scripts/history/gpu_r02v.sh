#!/bin/bash
# round 2, call V: deferred side launches x lane count
TAG=${1:-r02v}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
EXTRA="--wgrad-lanes 1" run defer1_l1 MH_DEFER_SIDE=1
EXTRA="--wgrad-lanes 1" run defer0_l1 MH_DEFER_SIDE=0
EXTRA="--wgrad-lanes 2" run defer0_l2 MH_DEFER_SIDE=0
EXTRA="--wgrad-lanes 1" run defer1_l1_tailnone MH_DEFER_SIDE=1 MH_PYR_TAIL_FLUSH=
EXTRA="--wgrad-lanes 1" run defer1_l1_group0 MH_DEFER_SIDE=1 MH_WGRAD_GROUP=0
EXTRA="--wgrad-lanes 1 --mode MAD" run mad_defer1_l1 MH_DEFER_SIDE=1
EXTRA="--wgrad-lanes 1 --precision bf16" run bf16_defer1_l1 MH_DEFER_SIDE=1
EXTRA="--wgrad-lanes 1 --precision fp32" run fp32_defer1_l1 MH_DEFER_SIDE=1
EXTRA="--wgrad-lanes 2 --precision fp32" run fp32_defer0_l2 MH_DEFER_SIDE=0
C="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 12 --warmup 3 --repeats 1 --wgrad-lanes 1"
(cd /tmp && MH_DEFER_SIDE=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_graph -o madnet -- python $GRAFT_REPO_ROOT/bench.py $C > $GRAFT_REPO_ROOT/$OUT/prof_graph.log 2>&1)
f=$(ls $OUT/prof_graph/*kernel_trace.csv | head -1)
python scripts/trace_timeline.py $f > $OUT/timeline.txt 2>&1
rm -rf $OUT/prof_graph
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e["config"].get("ops_per_step"))
PY
