#!/bin/bash
# round 3, call C: whole-step A/B of the streamed filter gradients (MH_WGRAD_STREAM=0 | 1) + per-kernel timeline of the replayed step + parity tests
TAG=${1:-r3c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-paths --no-cpu-baseline --no-roofline --no-step-surface"
for V in ${VARIANTS:-tiled stream}; do
  E="MH_X=0"; [ $V = tiled ] && E="MH_WGRAD_STREAM=0"; [ $V = nowgrad ] && E="MH_DEBUG_SKIP_WGRAD=1"; [ $V = serial ] && E="MH_WGRAD_LANES=0"
  env $E timeout 200 python bench.py $B --repeats 3 --steps 100 2>$OUT/bench_$V.err | tail -1 > $OUT/bench_$V.json
  (cd /tmp && env $E timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$V -o madnet -- python $GRAFT_REPO_ROOT/bench.py $B --steps 20 --warmup 5 --repeats 1 > $GRAFT_REPO_ROOT/$OUT/prof_$V.log 2>&1)
  f=$(find $OUT/prof_$V -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/trace_timeline.py $f > $OUT/timeline_$V.txt 2>&1
  g=$(find $OUT/prof_$V -name "*kernel_stats.csv" | head -1); [ -n "$g" ] && cp $g $OUT/kernel_stats_$V.csv; rm -rf $OUT/prof_$V
done
python - <<PY
import json
for v in "${VARIANTS:-tiled stream}".split():
    try:
        j = json.loads(open("$OUT/bench_%s.json" % v).read())
        print(v, j["ms_per_step"], "ms", j["value"], "pairs/s", "epe_vs_oracle", j.get("epe_vs_oracle"), "ws/grad", j.get("step_aggregate", {}).get("wgrad_ws_over_grad"))
    except Exception as e:
        print(v, "failed", e); print(open("$OUT/bench_%s.err" % v).read()[-1500:])
PY
for V in ${VARIANTS:-tiled stream}; do tail -4 $OUT/timeline_$V.txt; done
if [ -n "$TESTS" ]; then timeout 900 python -m pytest $TESTS -x -q -m gpu > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log; fi
