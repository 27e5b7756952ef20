#!/bin/bash
# The first GPU call of the next round (prepared at the end of round 3, when the GPU budget was spent):
#   1. the MI355X tests of the two prepared changes, which have never run on hardware (tests/test_zz_wgrad_image.py);
#   2. the whole-step A/B of that kernel (profiles/r03_experiments.txt #25: default plan vs 64 / 96 / 128 workgroups) and of the per-batch momentum update
#      (#26: engine.EARLY_UPDATE), alone and together, the base and the combination twice (box noise +-0.3 %);
#   3. the DispNet line (ragged-K iconv layers are in since #24).
# usage: gpurun --timeout 600 -- 'bash scripts/gpu_next_first.sh'
TAG=r4first; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 200 python -m pytest tests/test_zz_wgrad_image.py -m gpu -q 2>&1 | tail -5 | tee $OUT/pytest_wgrad_image.txt
SWEEP="base: img64:set=tune.wgrad_image=1 img96:set=tune.wgrad_image=96 img128:set=tune.wgrad_image=128 eu:set=engine.EARLY_UPDATE=True both:set=engine.EARLY_UPDATE=True,set=tune.wgrad_image=1 base2: both2:set=engine.EARLY_UPDATE=True,set=tune.wgrad_image=1" \
  bash scripts/gpu_sweep.sh $TAG 2>&1 | tee $OUT/sweep.txt
timeout 120 python bench.py --model dispnet --no-paths --no-cpu-baseline --no-roofline --no-step-surface --drift-steps 0 --repeats 3 2>/dev/null | tail -1 > $OUT/dispnet.json
