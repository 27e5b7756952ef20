#!/bin/bash
# round 6, call e: the collective through the C-ABI inside the step's graph -- tests + 1-rank bench lines (in-graph / host form / private)
OUT=gpurun_out/r6e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_distributed_gpu.py -m gpu -q -x 2>&1 | tail -15
Q="--no-paths --no-cpu-baseline --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 100 --repeats 3"
for v in "private:" "shared_graph:--shared-model" "shared_host:--shared-model --host-collective" "private2:" "shared_graph2:--shared-model" "mad_private:--mode MAD" "mad_shared_graph:--mode MAD --shared-model" "mad_shared_host:--mode MAD --shared-model --host-collective"; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 python bench.py $Q $f --detail e_$n.json 2>$OUT/$n.err | tail -1 > $OUT/$n.json
  python -c "import json;j=json.load(open('$OUT/$n.json'));print('$n', j['ms_per_step'], j.get('shared_model'), j.get('rccl'))" || tail -5 $OUT/$n.err
done
