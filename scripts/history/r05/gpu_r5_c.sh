#!/bin/bash
# round 5, third GPU call: the gather form of the row kernel -- parity, microbenchmark (graph-timed), A/B inside the replayed step
TAG=${1:-r5c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_parity.py -m gpu -q -x -k "corr" 2>&1 | tail -5 | tee $OUT/pytest_corr.txt
timeout 300 python scripts/exp/mb_corr_r05.py --no-d81 > $OUT/mb_corr.txt 2>&1; cat $OUT/mb_corr.txt
Q="--no-paths --no-cpu-baseline --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 20 --warmup 5 --repeats 3"
for v in 1 0 1 0; do
  timeout 300 python bench.py $Q --set tune.corr_row=$v 2>/dev/null | tail -1 > $OUT/bench_corr_row_$v.json
  python -c "import json;d=json.load(open('$OUT/bench_corr_row_$v.json'));print('corr_row=$v', d['ms_per_step'], d['timing']['ms_per_step_all'])"
done
timeout 600 python -m pytest tests/test_engine_parity.py tests/test_ref_graph.py -m gpu -q -x 2>&1 | tail -4 | tee $OUT/pytest_engine.txt
