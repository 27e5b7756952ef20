#!/bin/bash
# round 2, call B: parity after the surface / re-entrancy work, default bench line (x3 compile-time-K instances, uint8 input path),
# concurrent private streams per GPU, MAD line.
TAG=${1:-r02b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest_gpu.txt
timeout 900 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json
for S in 2 3 4 6; do
  timeout 300 python bench.py --concurrent-streams $S --steps 50 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cs$S.json
done
timeout 300 python bench.py --precision bf16 --concurrent-streams 4 --steps 50 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_bf16_cs4.json
timeout 300 python bench.py --mode MAD --steps 100 2>/dev/null | tail -1 > $OUT/bench_mad.json
tail -4 $OUT/pytest_gpu.txt; cut -c1-330 $OUT/bench_default.json; python - <<PY
import json
d=json.load(open("$OUT/bench_default.json"))
for k in ("epe_vs_oracle","roofline","roofline_dgrad","roofline_wgrad","step_surface","paths"):
    print(k, json.dumps(d.get(k))[:600])
for f in ("cs2","cs3","cs4","cs6","bf16_cs4","mad"):
    try:
        e=json.load(open("$OUT/bench_%s.json"%f)); print(f, e["value"], e["ms_per_step"])
    except Exception as ex: print(f, "ERR", ex)
PY
