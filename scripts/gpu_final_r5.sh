#!/bin/bash
# Round-5 artifact run: PMC passes first (bench.py reads profiles/r05_pmc_roofline.json), GPU test suite, the driver's bench line (+ tail stamps) and the variants,
# rocprofv3 kernel stats of the replayed MADNet FULL / MAD / DispNet steps, plan tables, microbenchmarks.  Raw profiler output stays in /tmp on the box;
# gpurun_out/$TAG gets the summaries.   SKIP_TESTS=1 / SKIP_PMC=1 / SKIP_VARIANTS=1 shorten it.
TAG=${1:-r5final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; WORK=/tmp/r5work; mkdir -p $WORK
R=$GRAFT_REPO_ROOT
if [ "$SKIP_PMC" != "1" ]; then
  bash scripts/gpu_pmc_r05.sh $TAG/pmc > $OUT/pmc.log 2>&1
  cp profiles/r05_pmc_roofline.json $OUT/r05_pmc_roofline.json 2>/dev/null
fi
if [ "$SKIP_TESTS" != "1" ]; then timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt; fi
timeout 900 python bench.py --steps 20 --warmup 5 --stamps 20 2>$OUT/bench.err | tail -1 > $OUT/bench_default_stamps.json
timeout 900 python bench.py --steps 20 --warmup 5 2>>$OUT/bench.err | tail -1 > $OUT/bench_default.json
if [ "$SKIP_VARIANTS" != "1" ]; then
  Q="--no-paths --no-cpu-baseline --no-roofline --no-step-surface --no-configs --drift-steps 0"
  timeout 300 python bench.py --mode MAD $Q 2>/dev/null | tail -1 > $OUT/bench_mad.json
  timeout 300 python bench.py --mode MAD --shared-model $Q 2>/dev/null | tail -1 > $OUT/bench_mad_shared_1gpu.json
  timeout 400 python bench.py --model dispnet --steps 30 $Q 2>/dev/null | tail -1 > $OUT/bench_dispnet_mixed.json
  timeout 300 python bench.py --concurrent-streams 4 --steps 30 $Q 2>/dev/null | tail -1 > $OUT/bench_private4.json
  timeout 300 python bench.py --streams-per-gpu 4 --steps 30 $Q 2>/dev/null | tail -1 > $OUT/bench_batched4.json
  timeout 300 python bench.py --shared-model --steps 30 $Q 2>/dev/null | tail -1 > $OUT/bench_shared_model_1gpu.json
  timeout 300 python bench.py --mode NONE --steps 30 $Q 2>/dev/null | tail -1 > $OUT/bench_none.json
fi
P="--steps 10 --warmup 2 --repeats 1 --min-region-seconds 0 --no-cpu-baseline --no-roofline --no-paths --no-step-surface --no-configs --drift-steps 0"
for v in "default:" "mad:--mode MAD" "dispnet:--model dispnet"; do
  n=${v%%:*}; f=${v#*:}
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $WORK/prof_$n -o $n -- python $R/bench.py $P $f > $R/$OUT/prof_$n.log 2>&1)
  k=$(find $WORK/prof_$n -name "*kernel_stats.csv" | head -1); [ -n "$k" ] && cp $k $OUT/bench_${n}_graph_kernel_stats.csv
  k=$(find $WORK/prof_$n -name "*kernel_trace.csv" | head -1); [ -n "$k" ] && python scripts/trace_timeline.py $k > $OUT/graph_timeline_$n.txt 2>&1
done
timeout 300 python scripts/plan_table.py > $OUT/plan_table_madnet.txt 2>&1
timeout 300 python scripts/plan_table.py --model dispnet > $OUT/plan_table_dispnet.txt 2>&1
timeout 300 python scripts/exp/mb_corr_r05.py > $OUT/microbench_corr.txt 2>&1
timeout 200 python scripts/exp/planes_phases_step.py > $OUT/planes_phases_step.txt 2>&1
python scripts/kernel_resources.py > $OUT/kernel_resources.txt 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -3 $OUT/smoke.txt
for f in $OUT/bench_*.json; do echo "$f: $(cut -c1-200 $f)"; done
tail -3 $OUT/graph_timeline_default.txt; head -3 $OUT/plan_table_madnet.txt; tail -5 $OUT/pmc.log; du -sh $OUT
