#!/bin/bash
# One gpurun call: GPU parity tests + smoke + default bench (bf16 + fp32 parity path) + MAD / fp32 / DispNet benches
# + rocprofv3 kernel stats of the default bench.   Usage (repo root on the GPU box): bash scripts/gpu_check.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > $OUT/env.log 2>&1
nproc >> $OUT/env.log
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.log 2>&1
echo "bench exit $?" >> $OUT/bench.log
timeout 300 python bench.py --precision fp32 --no-cpu-baseline > $OUT/bench_fp32.log 2>&1
timeout 300 python bench.py --mode MAD > $OUT/bench_mad.log 2>&1
timeout 300 python bench.py --model dispnet --precision fp32 --steps 20 > $OUT/bench_dispnet.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o madnet -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-parity-path > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1)
tail -4 $OUT/pytest.log; tail -2 $OUT/smoke.log
for f in bench bench_fp32 bench_mad bench_dispnet; do echo "== $f"; tail -1 $OUT/$f.log | cut -c1-2500; done
