#!/bin/bash
# round 6, call c: bias gradients without float atomics -- determinism probe, the new GPU tests, whole-step timing
OUT=gpurun_out/r6c; mkdir -p $OUT
timeout 300 python scripts/exp/det_probe.py > $OUT/det_probe_madnet.txt 2>&1; tail -8 $OUT/det_probe_madnet.txt
timeout 300 python scripts/exp/det_probe.py --model dispnet > $OUT/det_probe_dispnet.txt 2>&1; tail -8 $OUT/det_probe_dispnet.txt
timeout 900 python -m pytest tests/test_engine_parity.py tests/test_dispnet_parity.py tests/test_conv_parity.py tests/test_wgrad_stream.py tests/test_zz_wgrad_image.py tests/test_ops_parity.py -m gpu -q -x 2>&1 | tail -5
Q="--no-paths --no-cpu-baseline --no-roofline --no-step-surface --no-configs --drift-steps 0 --steps 100 --repeats 3"
for i in 1 2; do
  timeout 200 python bench.py $Q 2>/dev/null | tail -1 > $OUT/full_$i.json
  python -c "import json;j=json.load(open('$OUT/full_$i.json'));print('FULL', j['ms_per_step'], j['timing'])"
done
timeout 200 python bench.py $Q --model dispnet 2>/dev/null | tail -1 > $OUT/dispnet.json
python -c "import json;j=json.load(open('$OUT/dispnet.json'));print('dispnet', j['ms_per_step'], j['timing'])"
timeout 200 python bench.py $Q --mode MAD 2>/dev/null | tail -1 > $OUT/mad.json
python -c "import json;j=json.load(open('$OUT/mad.json'));print('MAD', j['ms_per_step'], j['timing'])"
