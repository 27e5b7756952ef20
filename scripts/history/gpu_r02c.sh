#!/bin/bash
# round 2, call C: fused level front end; x3 patch kernel phase breakdown; HW-queue experiment for concurrent streams;
# filter-gradient split cap sweep.
TAG=${1:-r02c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
timeout 600 python -m pytest tests/test_ops_parity.py tests/test_engine_parity.py tests/test_conv_parity.py -m gpu -x -q 2>&1 | tail -4 > $OUT/pytest_gpu.txt
timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/bench_fused.json
timeout 300 python scripts/microbench.py x3dbg 2>&1 | grep -v amdgpu.ids > $OUT/microbench_x3dbg.txt
for Q in 8 16; do
  GPU_MAX_HW_QUEUES=$Q timeout 300 python bench.py $B --concurrent-streams 4 2>/dev/null | tail -1 > $OUT/bench_cs4_q$Q.json
  GPU_MAX_HW_QUEUES=$Q timeout 300 python bench.py $B --concurrent-streams 4 --wgrad-lanes 0 2>/dev/null | tail -1 > $OUT/bench_cs4_q${Q}_nolanes.json
done
GPU_MAX_HW_QUEUES=16 timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/bench_cs1_q16.json
for S in 4 8 16 32 64; do
  MH_WGRAD_MAXSPLITS=$S timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/bench_maxsplits$S.json
done
cat $OUT/pytest_gpu.txt; cat $OUT/microbench_x3dbg.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        e=json.load(open(f)); print(f.split("/")[-1], "%.1f pairs/s  %.3f ms  ops %s ws %.0f MB"%(e["value"], e["ms_per_step"], e["config"].get("ops_per_step"), (e["step_aggregate"]["wgrad_ws_bytes"] or 0)/1e6))
    except Exception as ex: print(f, "ERR", ex)
PY
