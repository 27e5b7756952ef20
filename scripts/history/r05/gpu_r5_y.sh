#!/bin/bash
# round 5, call y: loss_tile_kernel with the three halo pixels of a thread loaded together (two dependent round trips instead of six): parity, the kernel alone, the steps
TAG=${1:-r5y}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_ops_parity.py tests/test_python_surface.py tests/test_ref_graph.py tests/test_engine_parity.py -q -m gpu -k "loss or reprojection or full_step or mad_step or ref_graph" 2>&1 | tail -2
timeout 300 python scripts/plan_table.py 2>/dev/null | grep -i "loss_tile" | head -3
Q="--steps 200 --warmup 20 --no-configs --no-cpu-baseline --no-paths --drift-steps 0 --no-step-surface --no-roofline"
for mode in FULL NONE MAD; do
  for i in 1 2; do
    timeout 300 python bench.py $Q --mode $mode 2>/dev/null | tail -1 > $OUT/bench_${mode}_$i.json
    python -c "import json; d=json.loads(open('$OUT/bench_${mode}_$i.json').read()); print('$mode #$i: %.4f ms/step' % d['ms_per_step'])"
  done
done
