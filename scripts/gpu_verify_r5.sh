#!/bin/bash
# round 5: what the driver runs at round end, in one call -- the GPU suite, smoke(), the default bench line
TAG=${1:-r5verify}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -3 $OUT/smoke.txt
timeout 600 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json
python - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read())
b=d["box"]
print("default: %.4f ms/step = %.1f pairs/s, epe %.3g, replay/launch-sum %.3f, power %s W" % (d["ms_per_step"], d["value"], d["epe_vs_oracle"], b["replay_over_launch_sum"], b["during_replay"]["power_w"]["median"]))
print({k: (round(v["ms_per_step"], 4), round(v["value"], 1)) for k, v in d["configs"].items()})
PY
