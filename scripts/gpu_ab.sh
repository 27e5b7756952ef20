#!/bin/bash
# tests + bench A/B over the conv M-tile + rocprof kernel stats (csv)
TAG=${1:-r01c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
timeout 150 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_default.log 2>&1
MH_CONV_BM=64 timeout 150 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_bm64.log 2>&1
MH_CONV_BM=128 timeout 150 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_bm128.log 2>&1
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o madnet -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-graph --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1)
tail -3 $OUT/pytest.log
for f in $OUT/bench_default.log $OUT/bench_bm64.log $OUT/bench_bm128.log; do echo "== $f"; grep -E "timed region|roofline" $f | cut -c1-200; tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('achieved'), d.get('roofline_corr',{}).get('achieved'))"; done
