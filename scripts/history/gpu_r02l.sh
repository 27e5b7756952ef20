#!/bin/bash
# round 2, call L: graph-replay TIMELINE of the default step (where do the 2.45 ms go: kernel time vs gaps on the main lane)
TAG=${1:-r02l}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
C="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 12 --warmup 3 --repeats 1"
for lanes in 2 0; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_graph_l$lanes -o madnet -- python $GRAFT_REPO_ROOT/bench.py $C --wgrad-lanes $lanes > $GRAFT_REPO_ROOT/$OUT/prof_graph_l$lanes.log 2>&1)
f=$(ls $OUT/prof_graph_l$lanes/*kernel_trace.csv | head -1)
python scripts/trace_timeline.py $f > $OUT/timeline_l$lanes.txt 2>&1
tail -4 $OUT/timeline_l$lanes.txt
rm -rf $OUT/prof_graph_l$lanes
done
timeout 300 python bench.py --no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3 2>/dev/null | tail -1 > $OUT/bench_default.json
python -c "
import json; e=json.load(open('$OUT/bench_default.json')); print(e['ms_per_step'], e['timing'])"
