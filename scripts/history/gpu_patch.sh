#!/bin/bash
# One gpurun call: A/B of the patch-staged bf16 conv kernel and of the MFMA register form (VGPR-form build = default .so,
# AGPR-form build = madnet_hip/libmadnet_hip_agpr.so when present), plus the conv / corr parity tests.
TAG=${1:-r09}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
AG=$PWD/real-time-self-adaptive-deep-stereo_amd/madnet_hip/libmadnet_hip_agpr.so
timeout 300 python -m pytest tests/test_conv_parity.py tests/test_ops_parity.py tests/test_engine_parity.py -m gpu -q -x --timeout 200 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
MH_CONV_PATCH=1 timeout 200 python -m pytest tests/test_engine_parity.py tests/test_dispnet_parity.py -m gpu -q -k "bf16" --timeout 150 -p no:cacheprovider > $OUT/pytest_patch.log 2>&1
echo "pytest(patch on) exit $?" >> $OUT/pytest_patch.log
timeout 150 python scripts/microbench.py patch > $OUT/micro_patch.log 2>&1
timeout 60 python scripts/microbench.py corr > $OUT/micro_corr.log 2>&1
B="python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-parity-path"
timeout 100 $B > $OUT/bench_vgpr.log 2>&1
MH_CONV_PATCH=1 timeout 100 $B > $OUT/bench_vgpr_patch.log 2>&1
timeout 100 $B --precision fp32 > $OUT/bench_vgpr_fp32.log 2>&1
if [ -f $AG ]; then
  MADNET_HIP_LIB=$AG timeout 100 $B > $OUT/bench_agpr.log 2>&1
  MADNET_HIP_LIB=$AG timeout 100 $B --precision fp32 > $OUT/bench_agpr_fp32.log 2>&1
  MADNET_HIP_LIB=$AG timeout 150 python scripts/microbench.py patch > $OUT/micro_patch_agpr.log 2>&1
fi
tail -3 $OUT/pytest.log; tail -3 $OUT/pytest_patch.log; cat $OUT/micro_patch.log | tail -26; tail -4 $OUT/micro_corr.log
for f in bench_vgpr bench_vgpr_patch bench_vgpr_fp32 bench_agpr bench_agpr_fp32; do echo "== $f $(tail -1 $OUT/$f.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],3))' 2>&1 | tail -1)"; done
