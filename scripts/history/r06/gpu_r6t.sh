#!/bin/bash
# round 6, call t: the prefetcher's new hand-over (no reader wake-ups from the consumer, no stream sync, event wait only when the upload is late, uint8 frames cast by the step's copy)
OUT=gpurun_out/r6t; mkdir -p $OUT
timeout 600 python -m pytest tests/test_api_gpu.py tests/test_cli_gpu.py tests/test_live_demo.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
timeout 500 python scripts/exp/prefetch_phases.py 300 2>&1 | grep -v "amdgpu.ids\|====\|WARNING\|MADNet ready" | tee $OUT/prefetch_phases.txt
timeout 500 python scripts/exp/step_surface_phases.py 300 2>&1 | grep -v "amdgpu.ids\|====\|WARNING\|MADNet ready" | tee $OUT/step_surface_phases.txt
timeout 300 python bench.py --no-paths --no-roofline --no-configs --drift-steps 0 --no-cpu-baseline 2>$OUT/bench.err | tail -1 > $OUT/bench.json
python -c "import json;j=json.load(open('$OUT/bench.json'));print('bench', j['ms_per_step'], j['value'], j['step_surface'])"
