#!/bin/bash
# Round-3 artifact run: GPU test suite, the driver's bench line + variants, rocprofv3 kernel stats (serial eager + graph replay + timeline of one
# replayed step), plan tables, PMC passes.  Raw profiler output stays in /tmp on the box; gpurun_out/$TAG gets the summaries (the merge back is
# limited to 64 MiB).
TAG=${1:-r3final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; WORK=/tmp/r3work; mkdir -p $WORK
if [ "$SKIP_TESTS" != "1" ]; then timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt; fi
timeout 600 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tail -1 > $OUT/bench_mixed.json
timeout 300 python bench.py --mode MAD 2>/dev/null | tail -1 > $OUT/bench_mad.json
timeout 300 python bench.py --model dispnet --steps 30 2>/dev/null | tail -1 > $OUT/bench_dispnet_mixed.json
timeout 300 python bench.py --streams-per-gpu 4 --steps 30 --no-paths --no-cpu-baseline --no-roofline --no-step-surface 2>/dev/null | tail -1 > $OUT/bench_batched4.json
timeout 300 python bench.py --streams-per-gpu 8 --steps 20 --no-paths --no-cpu-baseline --no-roofline --no-step-surface 2>/dev/null | tail -1 > $OUT/bench_batched8.json
timeout 300 python bench.py --shared-model --steps 30 --no-paths --no-cpu-baseline --no-roofline --no-step-surface 2>/dev/null | tail -1 > $OUT/bench_shared_model_1gpu.json
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $WORK/prof_serial -o madnet -- python $R/bench.py --steps 10 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-paths --no-step-surface --wgrad-lanes 0 > $R/$OUT/prof_serial.log 2>&1)
f=$(find $WORK/prof_serial -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/bench_eager_serial_kernel_stats_mixed.csv
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $WORK/prof_default -o madnet -- python $R/bench.py --steps 10 --warmup 2 --repeats 1 --min-region-seconds 0 --no-cpu-baseline --no-roofline --no-paths --no-step-surface > $R/$OUT/prof_default.log 2>&1)
f=$(find $WORK/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/bench_default_graph_kernel_stats.csv
f=$(find $WORK/prof_default -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/trace_timeline.py $f > $OUT/graph_timeline_default.txt 2>&1
timeout 300 python scripts/plan_table.py > $OUT/plan_table_madnet.txt 2>&1
timeout 300 python scripts/plan_table.py --model dispnet > $OUT/plan_table_dispnet.txt 2>&1
if [ "$SKIP_PMC" != "1" ]; then
  bash scripts/gpu_pmc_r03.sh $TAG/pmc > $OUT/pmc.log 2>&1
  cp profiles/r03_pmc_roofline.json $OUT/r03_pmc_roofline.json 2>/dev/null
  find $OUT/pmc -type f -size +3M -delete
fi
for f in $OUT/bench_*.json; do echo "$f: $(cut -c1-300 $f)"; done
tail -3 $OUT/graph_timeline_default.txt; head -3 $OUT/plan_table_madnet.txt; tail -5 $OUT/pmc.log; du -sh $OUT; du -sh gpurun_out
