#!/bin/bash
TAG=${1:-r01n}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_parity.py tests/test_abi.py tests/test_engine_parity.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
timeout 200 python bench.py --steps 40 --warmup 5 --precision bf16 > $OUT/bench_bf16.log 2>&1
timeout 200 python bench.py --steps 40 --warmup 5 > $OUT/bench_fp32.log 2>&1
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o madnet -- python $GRAFT_REPO_ROOT/bench.py --precision bf16 --steps 10 --warmup 2 --no-graph --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1)
for f in $OUT/bench_bf16.log $OUT/bench_fp32.log; do tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['dtype'], 'pairs/s', d['value'], 'ms', d['ms_per_step'], 'epe_vs_oracle', d.get('epe_vs_oracle'), 'conv TF', d.get('roofline',{}).get('achieved'))"; done
