#!/bin/bash
# A/B sweep of the whole step: SWEEP="name:ARG,ARG name2:..." where every ARG is a bench.py argument without its leading dashes, e.g.
#   SWEEP="base: norows:set=tune.conv_rows=0 nohead:set=engine.FUSE_HEAD=False lanes0:wgrad-lanes=0" scripts/gpu_sweep.sh tag
# (process environment: VAR=value items are still passed through `env` -- MH_CONV_PATCH, MH_CONV_BANK, MH_WGRAD_STREAM, MH_WGRAD_LANES,
#  MH_PYR_BF16_FROM are the switches the library / engines still read); prints ms/step per variant (3 x 100 steps, median)
TAG=${1:-sweep}; OUT=gpurun_out/$TAG; mkdir -p $OUT
B="--no-paths --no-cpu-baseline --no-roofline --no-step-surface ${BENCH_ARGS}"
for item in $SWEEP; do
  name=${item%%:*}; spec=${item#*:}; envs=""; extra=""
  for a in ${spec//,/ }; do
    case "$a" in
      [A-Z]*=*) envs="$envs $a" ;;
      *=*) extra="$extra --${a%%=*} ${a#*=}" ;;
      ?*) extra="$extra --$a" ;;
    esac
  done
  env $envs timeout 200 python bench.py $B --repeats 3 --steps 100 $extra 2>$OUT/$name.err | tail -1 > $OUT/$name.json
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$name.json").read())
    print("%-28s %.4f ms  %.1f pairs/s  min %.3f max %.3f" % ("$name", j["ms_per_step"], j["value"], j["timing"]["ms_per_step_min"], j["timing"]["ms_per_step_max"]))
except Exception as e:
    print("$name", "failed", e); print(open("$OUT/$name.err").read()[-800:])
PY
done
