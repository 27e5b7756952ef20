#!/bin/bash
# One gpurun call validating the D=81 correlation-gradient MFMA kernels: full GPU suite, corr microbench, DispNet benches.
TAG=${1:-r08}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
timeout 120 python scripts/microbench.py corr > $OUT/micro_corr.log 2>&1
timeout 200 python bench.py --model dispnet --steps 50 --no-cpu-baseline > $OUT/bench_dispnet.log 2>&1
timeout 200 python bench.py --model dispnet --precision fp32 --steps 30 --no-cpu-baseline > $OUT/bench_dispnet_fp32.log 2>&1
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log
tail -3 $OUT/pytest.log; tail -8 $OUT/micro_corr.log; tail -1 $OUT/smoke.log
for f in bench_dispnet bench_dispnet_fp32; do echo "== $f"; tail -1 $OUT/$f.log | cut -c1-1200; done
