#!/bin/bash
# round 5, first GPU call: parity of the new correlation kernels, their microbenchmark, the default bench line with the `configs` block, the GPU suite
TAG=${1:-r5a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_parity.py -m gpu -q -x -k "corr" 2>&1 | tail -5 | tee $OUT/pytest_corr.txt
timeout 300 python scripts/exp/mb_corr_r05.py > $OUT/mb_corr.txt 2>&1; cat $OUT/mb_corr.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json ) 2>&1 | grep real
grep "bench " $OUT/bench.err | tail -30
python - <<'PY'
import json,sys
d=json.loads(open("gpurun_out/%s/bench_default.json" % (sys.argv[1] if len(sys.argv)>1 else "r5a")).read())
print("headline %.4f ms  %.1f pairs/s  epe %s  roofline %.4f  box %s" % (d["ms_per_step"], d["value"], d.get("epe_vs_oracle"), d["roofline"].get("frac",-1), json.dumps(d.get("box"))[:400]))
for k,v in d.get("configs",{}).items():
    print(k, v.get("ms_per_step"), v.get("value"), v.get("epe_vs_oracle"), (v.get("roofline") or {}).get("kernel"), (v.get("roofline") or {}).get("frac"), v.get("cpu_baseline",{}).get("value"), v.get("error"))
for k in d:
    if k.startswith("roofline_corr"):
        print(k, d[k].get("kernel"), d[k].get("launch_ms"), d[k].get("frac"), d[k].get("error"), d[k].get("in_situ_B1_ms"), d[k].get("exact_fp32_frac"))
for e in d.get("kernel_families",[]):
    print("  %-70s n=%5.1f %7.1f us frac %s traffic %s" % (e["kernel"][:70], e["launches"], e["us_per_step"], e.get("frac"), e.get("traffic")))
PY
if [ "$SKIP_TESTS" != "1" ]; then timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt; fi
