#!/bin/bash
# round 3, call W: private-model streams of one GPU as parallel branches of ONE hipGraph (mh_plans_run) against one graph per stream
TAG=${1:-r3w}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_parity.py -m gpu -x -q -k "branches_of_one_graph" 2>&1 | tail -3
B="--no-paths --no-cpu-baseline --no-roofline --no-step-surface --repeats 3 --steps 50"
for cs in 2 3 4 6 8; do
  timeout 300 python bench.py $B --concurrent-streams $cs 2>$OUT/cs$cs.err | tail -1 > $OUT/cs$cs.json
done
timeout 300 python bench.py $B --concurrent-streams 4 --concurrent-graphs 2>$OUT/cg4.err | tail -1 > $OUT/cg4.json
timeout 300 python bench.py $B --concurrent-streams 4 --mode MAD 2>$OUT/mad4.err | tail -1 > $OUT/mad4.json
python - <<PY
import json
for n in ("cs2","cs3","cs4","cs6","cs8","cg4","mad4"):
    try:
        j=json.load(open("$OUT/%s.json"%n)); print(n, "%.1f pairs/s  %.3f ms per step of all streams" % (j["value"], j["ms_per_step"]), j["config"]["concurrent_private_streams_per_gpu"])
    except Exception as e:
        print(n, "failed", e); print(open("$OUT/%s.err"%n).read()[-1200:])
PY
