#!/bin/bash
# round 2, call F: full parity run, bench lines (default / fp32 / bf16 / MAD / DispNet x3 modes / batched), serial per-kernel profile of the
# mixed step, graph profile of the driver's command, correlation microbenchmarks by arithmetic mode.
TAG=${1:-r02f}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $OUT/smoke.txt
timeout 900 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json
timeout 300 python bench.py --mode MAD 2>/dev/null | tail -1 > $OUT/bench_mad.json
for P in mixed bf16 fp32; do
  timeout 300 python bench.py --model dispnet --precision $P --steps 30 --repeats 3 2>/dev/null | tail -1 > $OUT/bench_dispnet_$P.json
done
timeout 300 python bench.py --streams-per-gpu 4 --steps 30 --repeats 3 2>/dev/null | tail -1 > $OUT/bench_batched4.json
timeout 300 python bench.py --streams-per-gpu 8 --steps 20 --repeats 3 2>/dev/null | tail -1 > $OUT/bench_batched8.json
timeout 300 python bench.py --shared-model --steps 30 --repeats 3 2>/dev/null | tail -1 > $OUT/bench_shared_model_1rank.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_serial -o madnet -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --repeats 1 --no-graph --no-cpu-baseline --no-roofline --no-paths --no-step-surface --wgrad-lanes 0 > $GRAFT_REPO_ROOT/$OUT/prof_serial.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_default -o madnet -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-paths --no-step-surface > $GRAFT_REPO_ROOT/$OUT/prof_default.log 2>&1)
timeout 300 python scripts/microbench.py corr 2>&1 | grep -v amdgpu.ids > $OUT/microbench_corr.txt
timeout 300 python scripts/microbench.py x3dbg 2>&1 | grep -v amdgpu.ids > $OUT/microbench_x3dbg.txt
tail -3 $OUT/pytest_gpu.txt; cat $OUT/smoke.txt; cat $OUT/microbench_corr.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        e=json.load(open(f)); print(f.split("/")[-1], "%.1f pairs/s  %.3f ms  ops %s  epe %s"%(e["value"], e["ms_per_step"], e["config"].get("ops_per_step"), e.get("epe_vs_oracle")))
    except Exception as ex: print(f, "ERR", ex)
PY
