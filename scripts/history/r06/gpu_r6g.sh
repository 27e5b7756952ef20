#!/bin/bash
# round 6, call g: the whole GPU suite + smoke on the current tree (log kept whole)
OUT=gpurun_out/r6g; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.txt
