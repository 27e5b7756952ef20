#!/bin/bash
# round 6, call b: determinism probe of the product path + placement modes of the small-layer kernel (microbench + whole-step A/B)
OUT=gpurun_out/r6b; mkdir -p $OUT
timeout 300 python scripts/exp/det_probe.py > $OUT/det_probe_madnet.txt 2>&1; tail -40 $OUT/det_probe_madnet.txt
timeout 300 python scripts/exp/det_probe.py --model dispnet > $OUT/det_probe_dispnet.txt 2>&1; tail -30 $OUT/det_probe_dispnet.txt
timeout 600 python -m pytest tests/test_conv_parity.py -m gpu -q -x -k "placement or small_layer" 2>&1 | tail -3
timeout 600 python scripts/exp/mb_small_place.py > $OUT/mb_small_place.txt 2>&1; cat $OUT/mb_small_place.txt
Q="--no-paths --no-cpu-baseline --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 100 --repeats 3"
for m in 0 3 1 2 0 3; do
  timeout 200 python bench.py $Q --set tune.conv_bank_small=$m 2>/dev/null | tail -1 > $OUT/ab_$m.json
  python -c "import json;j=json.load(open('$OUT/ab_$m.json'));print('mode $m', j['ms_per_step'], j['timing'])"
done
