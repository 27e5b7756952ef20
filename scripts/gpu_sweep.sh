#!/bin/bash
# environment-variable sweep of the whole step: SWEEP="name:VAR=val,VAR2=val2 name2:..." ; prints ms/step per variant (3 x 100 steps, median)
TAG=${1:-sweep}; OUT=gpurun_out/$TAG; mkdir -p $OUT
B="--no-paths --no-cpu-baseline --no-roofline --no-step-surface ${BENCH_ARGS}"
for item in $SWEEP; do
  name=${item%%:*}; envs=${item#*:}; envs=${envs//,/ }
  env $envs timeout 200 python bench.py $B --repeats 3 --steps 100 2>$OUT/$name.err | tail -1 > $OUT/$name.json
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$name.json").read())
    print("%-28s %.4f ms  %.1f pairs/s  all %s" % ("$name", j["ms_per_step"], j["value"], ["%.3f" % x for x in j["timing"]["ms_per_step_all"]]))
except Exception as e:
    print("$name", "failed", e); print(open("$OUT/$name.err").read()[-800:])
PY
done
