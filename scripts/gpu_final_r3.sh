#!/bin/bash
# Round-3 artifact run: GPU test suite, the driver's bench line + variants, rocprofv3 kernel stats (serial eager + graph replay + timeline of one
# replayed step), plan tables, PMC passes.  Outputs under gpurun_out/$TAG (the summaries are copied to profiles/ by hand).
TAG=${1:-r3final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ "$SKIP_TESTS" != "1" ]; then timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt; fi
timeout 600 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tail -1 > $OUT/bench_mixed.json
timeout 300 python bench.py --mode MAD 2>/dev/null | tail -1 > $OUT/bench_mad.json
timeout 300 python bench.py --model dispnet --steps 30 2>/dev/null | tail -1 > $OUT/bench_dispnet_mixed.json
timeout 300 python bench.py --streams-per-gpu 4 --steps 30 --no-paths --no-cpu-baseline --no-roofline --no-step-surface 2>/dev/null | tail -1 > $OUT/bench_batched4.json
timeout 300 python bench.py --streams-per-gpu 8 --steps 20 --no-paths --no-cpu-baseline --no-roofline --no-step-surface 2>/dev/null | tail -1 > $OUT/bench_batched8.json
timeout 300 python bench.py --shared-model --steps 30 --no-paths --no-cpu-baseline --no-roofline --no-step-surface 2>/dev/null | tail -1 > $OUT/bench_shared_model_1gpu.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_serial -o madnet -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-paths --no-step-surface --wgrad-lanes 0 > $GRAFT_REPO_ROOT/$OUT/prof_serial.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_default -o madnet -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --repeats 1 --min-region-seconds 0 --no-cpu-baseline --no-roofline --no-paths --no-step-surface > $GRAFT_REPO_ROOT/$OUT/prof_default.log 2>&1)
T=$(ls $OUT/prof_default/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$T" ] && python scripts/trace_timeline.py $T > $OUT/graph_timeline_default.txt 2>&1
timeout 300 python scripts/plan_table.py > $OUT/plan_table_madnet.txt 2>&1
timeout 300 python scripts/plan_table.py --model dispnet > $OUT/plan_table_dispnet.txt 2>&1
if [ "$SKIP_PMC" != "1" ]; then bash scripts/gpu_pmc_r03.sh $TAG/pmc > $OUT/pmc.log 2>&1; fi
for f in $OUT/bench_*.json; do echo "$f: $(cut -c1-300 $f)"; done
tail -3 $OUT/graph_timeline_default.txt; head -3 $OUT/plan_table_madnet.txt
