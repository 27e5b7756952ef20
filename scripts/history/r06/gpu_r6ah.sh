#!/bin/bash
# round 6, call ah: NODEFER no longer honoured on lane-to-lane join ops: the cut update probe again, the shared-model step (collective at the cut), distributed + engine tests
OUT=gpurun_out/r6ah; mkdir -p $OUT
timeout 800 python scripts/exp/cut_update_probe.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee $OUT/cut_update_probe.txt
timeout 1200 python -m pytest tests/test_distributed_gpu.py tests/test_api_gpu.py tests/test_engine_parity.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
Q="--no-paths --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 50 --repeats 3 --no-cpu-baseline"
for v in "private:" "shared:--shared-model" "private2:" "shared2:--shared-model" "mad_shared:--mode MAD --shared-model"; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail ah_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j['value'], (j.get('shared_model') or {}).get('collective_ms_in_step'))" || tail -3 $OUT/$n.err
done
