#!/bin/bash
# Artifact run of a round (ROUND=r06 by default): PMC passes first (bench.py reads profiles/${ROUND}_pmc_roofline.json), GPU test suite, the driver's bench command
# (compact line + bench_detail.json) and the variants, rocprofv3 kernel stats of the replayed MADNet FULL / MAD / DispNet steps, plan tables, microbenchmarks.
# Raw profiler output stays in /tmp on the box; gpurun_out/$TAG gets the summaries; scripts/collect_profiles.sh copies them into profiles/${ROUND}_*.
#   bash scripts/gpu_final.sh <tag>        SKIP_TESTS=1 / SKIP_PMC=1 / SKIP_VARIANTS=1 / SKIP_PROF=1 shorten it.
export ROUND=${ROUND:-r06}
TAG=${1:-${ROUND}final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; WORK=/tmp/${ROUND}work; mkdir -p $WORK
R=$GRAFT_REPO_ROOT
bench() {  # bench <name> <args...>: the line -> $OUT/bench_<name>.json, the complete record -> $OUT/bench_<name>_detail.json
  local n=$1; shift
  timeout 900 python bench.py "$@" --detail bench_${n}_detail.json 2>>$OUT/bench.err | tail -1 > $OUT/bench_$n.json
  mv -f bench_${n}_detail.json $OUT/ 2>/dev/null; rm -f gpurun_out/bench_${n}_detail.json
}
if [ "$SKIP_PMC" != "1" ]; then
  bash scripts/gpu_pmc.sh $TAG/pmc > $OUT/pmc.log 2>&1
  cp profiles/${ROUND}_pmc_roofline.json $OUT/ 2>/dev/null
fi
if [ "$SKIP_TESTS" != "1" ]; then   # (RCCL prints a version banner through C stdio when the test processes exit: keep the log whole, quote the summary line)
  timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $OUT/pytest_gpu.txt
  grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3 >> $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -3 $OUT/smoke.txt
# the driver's own command, stdout kept whole (the ONE line, < 6 KB)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2>>$OUT/bench.err; cp bench_detail.json $OUT/bench_default_detail.json
wc -c $OUT/bench_default.json
bench default_stamps --steps 20 --warmup 5 --stamps 20
if [ "$SKIP_VARIANTS" != "1" ]; then
  Q="--no-paths --no-cpu-baseline --no-roofline --no-step-surface --no-configs --drift-steps 0"
  bench mad --mode MAD $Q
  bench mad_shared_1gpu --mode MAD --shared-model $Q
  bench dispnet_mixed --model dispnet --steps 30 $Q
  bench private4 --concurrent-streams 4 --steps 30 $Q
  bench batched4 --streams-per-gpu 4 --steps 30 $Q
  bench shared_model_1gpu --shared-model --steps 30 $Q
  bench none --mode NONE --steps 30 $Q
fi
if [ "$SKIP_PROF" != "1" ]; then
  P="--steps 10 --warmup 2 --repeats 1 --min-region-seconds 0 --no-cpu-baseline --no-roofline --no-paths --no-step-surface --no-configs --drift-steps 0"
  for v in "default:" "mad:--mode MAD" "dispnet:--model dispnet"; do
    n=${v%%:*}; f=${v#*:}
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $WORK/prof_$n -o $n -- python $R/bench.py $P $f > $R/$OUT/prof_$n.log 2>&1)
    k=$(find $WORK/prof_$n -name "*kernel_stats.csv" | head -1); [ -n "$k" ] && cp $k $OUT/bench_${n}_graph_kernel_stats.csv
    k=$(find $WORK/prof_$n -name "*kernel_trace.csv" | head -1); [ -n "$k" ] && python scripts/trace_timeline.py $k > $OUT/graph_timeline_$n.txt 2>&1
  done
  timeout 300 python scripts/plan_table.py > $OUT/plan_table_madnet.txt 2>&1
  timeout 300 python scripts/plan_table.py --model dispnet > $OUT/plan_table_dispnet.txt 2>&1
  timeout 300 python scripts/exp/mb_corr.py > $OUT/microbench_corr.txt 2>&1
  timeout 200 python scripts/exp/planes_phases_step.py > $OUT/planes_phases_step.txt 2>&1
  python scripts/kernel_resources.py > $OUT/kernel_resources.txt 2>/dev/null
fi
for f in $OUT/bench_*.json; do case $f in *_detail.json) ;; *) echo "$f: $(cut -c1-160 $f)";; esac; done
tail -3 $OUT/graph_timeline_default.txt 2>/dev/null; head -3 $OUT/plan_table_madnet.txt 2>/dev/null; tail -5 $OUT/pmc.log 2>/dev/null; du -sh $OUT
