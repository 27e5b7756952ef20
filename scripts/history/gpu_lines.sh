#!/bin/bash
# Extra bench lines (no cpu baseline / roofline): batched streams, MAD, fp32 -- for profiles/.
TAG=${1:-lines}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python bench.py --streams-per-gpu 4 --steps 30 --no-cpu-baseline --no-roofline --no-parity-path 2>/dev/null | tail -1 > $OUT/bench_batched4.json
timeout 60 python bench.py --streams-per-gpu 8 --steps 20 --no-cpu-baseline --no-roofline --no-parity-path 2>/dev/null | tail -1 > $OUT/bench_batched8.json
timeout 60 python bench.py --mode MAD --no-cpu-baseline --no-roofline --no-parity-path 2>/dev/null | tail -1 > $OUT/bench_mad.json
for f in $OUT/bench_*.json; do echo "$f: $(python -c 'import sys,json; d=json.load(open(sys.argv[1])); print(round(d["value"],1), round(d["ms_per_step"],3))' $f)"; done
