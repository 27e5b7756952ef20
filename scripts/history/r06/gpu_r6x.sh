#!/bin/bash
# round 6, call x: the step's frames through the input table (mh_fetch_inputs as the plan's first op): parity + the reference-FPS loop and MAD through Adapter.step
OUT=gpurun_out/r6x; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_api_gpu.py tests/test_cli_gpu.py tests/test_distributed_gpu.py tests/test_ops_parity.py tests/test_live_demo.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
timeout 500 python scripts/exp/step_surface_phases.py 300 2>&1 | grep -v "amdgpu.ids\|====\|WARNING\|MADNet ready" | tee $OUT/step_surface_phases.txt
timeout 300 python bench.py --no-paths --no-roofline --no-configs --drift-steps 0 --no-cpu-baseline 2>$OUT/bench.err | tail -1 > $OUT/bench.json
python -c "import json;j=json.load(open('$OUT/bench.json'));print('bench', j['ms_per_step'], j['value'], j['step_surface'])"
for r in 1 2; do
timeout 300 python bench.py --mode MAD --no-paths --no-roofline --no-configs --drift-steps 0 --no-cpu-baseline --no-step-surface 2>$OUT/mad$r.err | tail -1 > $OUT/mad$r.json
python -c "import json;j=json.load(open('$OUT/mad$r.json'));print('mad', j['ms_per_step'], j['value'])"
done
