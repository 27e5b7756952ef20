#!/bin/bash
# round 2, last call: all-taps filter-gradient kernel -- microbenchmark + whole-step A/B (default vs MH_WGRAD_TAPS=1)
mkdir -p gpurun_out
timeout 60 python scripts/microbench.py wgradt > gpurun_out/r04d_wgradt.txt 2>&1
B="--no-paths --no-cpu-baseline --no-roofline --no-step-surface --repeats 3 --steps 100"
timeout 60 python bench.py $B > gpurun_out/r04d_bench_default.json 2> gpurun_out/r04d_bench_default.err
MH_WGRAD_TAPS=1 timeout 60 python bench.py $B > gpurun_out/r04d_bench_taps.json 2> gpurun_out/r04d_bench_taps.err
cat gpurun_out/r04d_wgradt.txt | cut -c1-300
python - <<'PY'
import json
for n in ("default", "taps"):
    try:
        j = json.loads(open("gpurun_out/r04d_bench_%s.json" % n).read().strip().splitlines()[-1])
        print(n, j["ms_per_step"], j["value"], j.get("epe_vs_oracle"), j.get("config", {}).get("launches"))
    except Exception as e:
        print(n, "failed", e, open("gpurun_out/r04d_bench_%s.err" % n).read()[-600:])
PY
