#!/bin/bash
# Short gpurun call: patch-kernel parity tests, its microbench, and the whole-step bench with the kernel off / on.
TAG=${1:-r10}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_conv_parity.py -m gpu -q -x -k "patch or bf16" --timeout 150 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
timeout 150 python scripts/microbench.py patch > $OUT/micro_patch.log 2>&1
B="python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-parity-path"
MH_CONV_PATCH=0 timeout 100 $B > $OUT/bench_off.log 2>&1
MH_CONV_PATCH=1 timeout 100 $B > $OUT/bench_on.log 2>&1
MH_CONV_PATCH=1 timeout 200 python -m pytest tests/test_engine_parity.py -m gpu -q -k "bf16" --timeout 150 -p no:cacheprovider > $OUT/pytest_on.log 2>&1
echo "pytest(patch on) exit $?" >> $OUT/pytest_on.log
tail -2 $OUT/pytest.log; tail -2 $OUT/pytest_on.log; tail -26 $OUT/micro_patch.log
for f in bench_off bench_on; do echo "== $f $(tail -1 $OUT/$f.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],3), d["config"].get("final_loss"))' 2>&1 | tail -1)"; done
