#!/bin/bash
# round 5: small-layer bank kernel with one-round-trip prologue -- parity, step time, plan table
TAG=${1:-r5g}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_parity.py tests/test_ops_parity.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest_conv.txt
Q="--no-paths --no-cpu-baseline --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 20 --warmup 5 --repeats 3"
for i in 1 2; do
  timeout 300 python bench.py $Q 2>/dev/null | tail -1 > $OUT/bench_q_$i.json
  python -c "import json;d=json.load(open('$OUT/bench_q_$i.json'));print('FULL', d['ms_per_step'], d['timing']['ms_per_step_all'])"
done
timeout 300 python bench.py $Q --mode MAD 2>/dev/null | tail -1 > $OUT/bench_mad.json; python -c "import json;d=json.load(open('$OUT/bench_mad.json'));print('MAD', d['ms_per_step'])"
timeout 300 python bench.py $Q --mode NONE 2>/dev/null | tail -1 > $OUT/bench_none.json; python -c "import json;d=json.load(open('$OUT/bench_none.json'));print('NONE', d['ms_per_step'])"
timeout 300 python scripts/plan_table.py > $OUT/plan_table_madnet.txt 2>&1; head -40 $OUT/plan_table_madnet.txt
timeout 900 python -m pytest tests/test_engine_parity.py tests/test_ref_graph.py tests/test_dispnet_parity.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest_engine.txt
