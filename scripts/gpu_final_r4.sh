#!/bin/bash
# Round-4 artifact run: GPU test suite, the driver's bench line (+ tail stamps) and the variants, rocprofv3 kernel stats / timeline of the replayed step,
# plan tables, microbenchmarks, PMC passes (L2 flushed before every measured launch).  Raw profiler output stays in /tmp on the box; gpurun_out/$TAG gets
# the summaries (the merge back is limited to 64 MiB).   SKIP_TESTS=1 / SKIP_PMC=1 / SKIP_VARIANTS=1 shorten it.
TAG=${1:-r4final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; WORK=/tmp/r4work; mkdir -p $WORK
R=$GRAFT_REPO_ROOT
if [ "$SKIP_TESTS" != "1" ]; then timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt; fi
timeout 900 python bench.py --stamps 20 2>$OUT/bench.err | tail -1 > $OUT/bench_default_stamps.json
timeout 900 python bench.py 2>>$OUT/bench.err | tail -1 > $OUT/bench_default.json
if [ "$SKIP_VARIANTS" != "1" ]; then
  Q="--no-paths --no-cpu-baseline --no-roofline --no-step-surface --drift-steps 0"
  timeout 300 python bench.py --mode MAD 2>/dev/null | tail -1 > $OUT/bench_mad.json
  timeout 300 python bench.py --mode MAD --shared-model $Q 2>/dev/null | tail -1 > $OUT/bench_mad_shared_1gpu.json
  timeout 400 python bench.py --model dispnet --steps 30 2>/dev/null | tail -1 > $OUT/bench_dispnet_mixed.json
  timeout 300 python bench.py --concurrent-streams 4 --steps 30 $Q 2>/dev/null | tail -1 > $OUT/bench_private4.json
  timeout 300 python bench.py --streams-per-gpu 4 --steps 30 $Q 2>/dev/null | tail -1 > $OUT/bench_batched4.json
  timeout 300 python bench.py --shared-model --steps 30 $Q 2>/dev/null | tail -1 > $OUT/bench_shared_model_1gpu.json
  timeout 300 python bench.py --mode NONE --steps 30 $Q 2>/dev/null | tail -1 > $OUT/bench_none.json
fi
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $WORK/prof_default -o madnet -- python $R/bench.py --steps 10 --warmup 2 --repeats 1 --min-region-seconds 0 --no-cpu-baseline --no-roofline --no-paths --no-step-surface --drift-steps 0 > $R/$OUT/prof_default.log 2>&1)
f=$(find $WORK/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/bench_default_graph_kernel_stats.csv
f=$(find $WORK/prof_default -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/trace_timeline.py $f > $OUT/graph_timeline_default.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $WORK/prof_dispnet -o dispnet -- python $R/bench.py --model dispnet --steps 10 --warmup 2 --repeats 1 --min-region-seconds 0 --no-cpu-baseline --no-roofline --no-paths --no-step-surface --drift-steps 0 > $R/$OUT/prof_dispnet.log 2>&1)
f=$(find $WORK/prof_dispnet -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/bench_dispnet_graph_kernel_stats.csv
f=$(find $WORK/prof_dispnet -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/trace_timeline.py $f > $OUT/graph_timeline_dispnet.txt 2>&1
timeout 300 python scripts/plan_table.py > $OUT/plan_table_madnet.txt 2>&1
timeout 300 python scripts/plan_table.py --model dispnet > $OUT/plan_table_dispnet.txt 2>&1
timeout 300 python scripts/microbench.py planes > $OUT/microbench_planes.txt 2>&1
timeout 300 python scripts/microbench.py dispnet > $OUT/microbench_dispnet.txt 2>&1
timeout 200 python scripts/exp/node_floor.py > $OUT/node_floor.txt 2>&1
if [ "$SKIP_PMC" != "1" ]; then
  bash scripts/gpu_pmc_r04.sh $TAG/pmc > $OUT/pmc.log 2>&1
  cp profiles/r04_pmc_roofline.json $OUT/r04_pmc_roofline.json 2>/dev/null
  find $OUT/pmc -type f -size +3M -delete
fi
for f in $OUT/bench_*.json; do echo "$f: $(cut -c1-260 $f)"; done
tail -3 $OUT/graph_timeline_default.txt; head -3 $OUT/plan_table_madnet.txt; tail -5 $OUT/pmc.log; du -sh $OUT
