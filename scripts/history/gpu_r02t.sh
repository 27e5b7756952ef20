#!/bin/bash
# round 2, call T: small-layer fragment-bank kernel (forward + input gradient of the 1/16-1/64 levels): parity, A/B, timeline
TAG=${1:-r02t}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
timeout 900 python -m pytest tests/test_conv_parity.py tests/test_engine_parity.py -m gpu -x -q 2>&1 | tail -5 > $OUT/pytest_gpu.txt
cat $OUT/pytest_gpu.txt
run small4096 MH_X=1
run small0 MH_CONV_BANK_SMALL_MAXPIX=0
run small2048 MH_CONV_BANK_SMALL_MAXPIX=2048
run small8192 MH_CONV_BANK_SMALL_MAXPIX=8192
run bank0 MH_CONV_BANK=0
EXTRA="--precision bf16" run bf16_small4096 MH_X=1
EXTRA="--precision bf16" run bf16_bank0 MH_CONV_BANK=0
EXTRA="--mode MAD" run mad_small4096 MH_X=1
EXTRA="--mode MAD" run mad_bank0 MH_CONV_BANK=0
run small4096_again MH_X=1
C="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 12 --warmup 3 --repeats 1"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_graph -o madnet -- python $GRAFT_REPO_ROOT/bench.py $C > $GRAFT_REPO_ROOT/$OUT/prof_graph.log 2>&1)
f=$(ls $OUT/prof_graph/*kernel_trace.csv | head -1)
python scripts/trace_timeline.py $f > $OUT/timeline.txt 2>&1
rm -rf $OUT/prof_graph
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]], e["config"].get("ops_per_step"))
PY
