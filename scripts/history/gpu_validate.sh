#!/bin/bash
# One gpurun call: the full GPU suite + smoke + the default bench line + the DispNet line (what the driver runs at round end).
TAG=${1:-val}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
timeout 90 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log
timeout 400 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_bf16.json
timeout 200 python bench.py --model dispnet --steps 50 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_dispnet_bf16.json
tail -3 $OUT/pytest.log; tail -2 $OUT/smoke.log
for f in $OUT/bench_*.json; do echo "$f: $(cut -c1-1800 $f)"; done
