#!/bin/bash
# round 6, call f: first-writer back end + lane-to-lane join -- tests of the touched paths, A/B of the step, shared-model lines
OUT=gpurun_out/r6f; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_engine_parity.py tests/test_ops_parity.py tests/test_distributed_gpu.py tests/test_ref_graph.py tests/test_api_gpu.py -m gpu -q -x 2>&1 | tail -5
timeout 300 python scripts/exp/det_probe.py 2>&1 | tail -5
Q="--no-paths --no-cpu-baseline --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 100 --repeats 3"
for v in "fw_on:" "fw_off:--set engine.FIRST_WRITER=False" "fw_on2:" "fw_off2:--set engine.FIRST_WRITER=False" "shared_graph:--shared-model" "shared_host:--shared-model --host-collective" "fw_on3:" "shared_graph2:--shared-model" "mad:--mode MAD" "mad_shared:--mode MAD --shared-model" "mad_full_cfg:--mode MAD --block-config MadNet_full.json" "mad_full_cfg_off:--mode MAD --block-config MadNet_full.json --set engine.FIRST_WRITER=False"; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail f_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j['config'].get('ops_per_step'), (j.get('shared_model') or {}).get('collective_ms_in_step'))" || tail -5 $OUT/$n.err
done
