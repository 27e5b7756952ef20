#!/bin/bash
# microbench (optional) + bench x2 + rocprof kernel stats
TAG=${1:-r01f}; MICRO=${2:-conv}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ "$MICRO" != "none" ]; then timeout 400 python scripts/microbench.py $MICRO > $OUT/micro.log 2>&1; fi
timeout 100 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench1.log 2>&1
timeout 100 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench2.log 2>&1
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o madnet -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-graph --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1)
for f in $OUT/bench1.log $OUT/bench2.log; do tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('pairs/s', d['value'], 'ms', d['ms_per_step'], 'conv TF', d.get('roofline',{}).get('achieved'), 'corr GB/s', d.get('roofline_corr',{}).get('achieved'))"; done
