#!/bin/bash
# round 3, call K: K = 1 input-gradient kernel of the disparity heads, two-piece shared-model all-reduce on a 1-rank RCCL group, A/B of the
# level-3 input gradients (small-layer bank kernel / patch kernel instead of the tiled one) and of the streamed kernel's workgroup target
TAG=${1:-r3k}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_parity.py -m gpu -x -q -k "k1_dgrad or dgrad_wgrad" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_distributed_gpu.py -m gpu -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_engine_parity.py tests/test_api_gpu.py -m gpu -x -q 2>&1 | tail -3
SWEEP="base:MH_X=0 l3bank:MH_CONV_BANK_SMALL_MAXPIX_DGRAD=8192 l3patch:MH_CONV_PATCH_MINPIX=7680 wgs224:MH_WGRAD_STREAM_WGS=224 wgs192:MH_WGRAD_STREAM_WGS=192 base2:MH_X=0" bash scripts/gpu_sweep.sh $TAG
timeout 300 python bench.py --shared-model --steps 50 --repeats 3 --no-paths --no-cpu-baseline --no-roofline --no-step-surface 2>$OUT/shared.err | tail -1 > $OUT/shared.json
timeout 300 python bench.py --shared-model --late-reduce --steps 50 --repeats 3 --no-paths --no-cpu-baseline --no-roofline --no-step-surface 2>$OUT/shared_late.err | tail -1 > $OUT/shared_late.json
python - <<PY
import json
for n in ("shared", "shared_late"):
    try:
        j = json.load(open("$OUT/%s.json" % n)); print(n, "%.4f ms" % j["ms_per_step"], j.get("shared_model"))
    except Exception as e:
        print(n, "failed", e); print(open("$OUT/%s.err" % n).read()[-1500:])
PY
timeout 300 python scripts/plan_table.py > $OUT/plan_table_madnet.txt 2>&1; head -30 $OUT/plan_table_madnet.txt
