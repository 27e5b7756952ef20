#!/bin/bash
TAG=${1:-r02e}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/bench_fused1.json
MH_FUSE_FRONT=0 timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/bench_fused0.json
timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/bench_fused1b.json
MH_FUSE_FRONT=0 timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/bench_fused0b.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_graph -o madnet -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-roofline --no-paths --no-step-surface > $GRAFT_REPO_ROOT/$OUT/prof_graph.log 2>&1)
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], e["timing"]["ms_per_step_all"], e["config"]["ops_per_step"])
PY
head -12 $OUT/prof_graph/madnet_kernel_stats.csv | cut -c1-200
grep -i "level_front\|corr_fwd\|warp_fwd\|resize_fwd" $OUT/prof_graph/madnet_kernel_stats.csv | cut -c1-220
