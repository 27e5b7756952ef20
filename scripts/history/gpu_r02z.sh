#!/bin/bash
# round 2, call Z: with deferred side launches, re-test the small critical-path kernels on side lanes (loss value / metrics; warp-gradient scatters)
TAG=${1:-r02z}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
timeout 600 python -m pytest tests/test_conv_parity.py -m gpu -x -q -k "bank" 2>&1 | tail -2
run base MH_X=1
run sideloss MH_SIDE_LOSS=1
run scatter1 MH_SCATTER_LANE=1
run scatter2 MH_SCATTER_LANE=2
run sideloss_scatter2 MH_SIDE_LOSS=1 MH_SCATTER_LANE=2
run lanes2_sideloss MH_SIDE_LOSS=1 MH_WGRAD_LANES=2
run base_again MH_X=1
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e["config"].get("ops_per_step"))
PY
