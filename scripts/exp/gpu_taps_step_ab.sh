#!/bin/bash
# Whole-step A/B of the all-taps filter-gradient kernel with per-kernel timelines of one replayed step (default vs MH_WGRAD_TAPS=1): where the step gets
# slower although the kernel is faster stand-alone (profiles/r02_microbench_wgrad_taps.txt).  ~1.5 GPU-minutes.
TAG=${1:-taps_ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-paths --no-cpu-baseline --no-roofline --no-step-surface"
for V in default taps; do
  E=""; [ $V = taps ] && E="MH_WGRAD_TAPS=1"
  env $E timeout 120 python bench.py $B --repeats 3 --steps 100 2>/dev/null | tail -1 > $OUT/bench_$V.json
  (cd /tmp && env $E timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$V -o madnet -- python $GRAFT_REPO_ROOT/bench.py $B --steps 20 --warmup 5 --repeats 1 > $GRAFT_REPO_ROOT/$OUT/prof_$V.log 2>&1)
  f=$(ls $OUT/prof_$V/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && python scripts/trace_timeline.py $f > $OUT/timeline_$V.txt 2>&1
  cp $OUT/prof_$V/*kernel_stats.csv $OUT/kernel_stats_$V.csv 2>/dev/null; rm -rf $OUT/prof_$V
done
python - <<PY
import json
for v in ("default", "taps"):
    try:
        j = json.loads(open("$OUT/bench_%s.json" % v).read())
        print(v, j["ms_per_step"], "ms", j["value"], "pairs/s")
    except Exception as e:
        print(v, "failed", e)
PY
tail -3 $OUT/timeline_default.txt $OUT/timeline_taps.txt
