#!/bin/bash
# One gpurun call: GPU parity tests + bench (eager & hipGraph) + rocprofv3 kernel stats.
# Usage (from the repo root on the GPU box): bash scripts/gpu_check.sh [tag]
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
timeout 300 python bench.py --steps 20 --warmup 3 --no-graph --no-cpu-baseline --no-roofline > $OUT/bench_eager.log 2>&1
timeout 600 python bench.py --steps 50 --warmup 5 > $OUT/bench.log 2>&1
echo "bench exit $?" >> $OUT/bench.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o madnet -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-graph --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1)
ls -R $OUT/prof 2>/dev/null | head -20 >> $OUT/prof.log
tail -5 $OUT/pytest.log; cat $OUT/smoke.log | tail -3; cat $OUT/bench_eager.log | tail -2; cat $OUT/bench.log | tail -2
