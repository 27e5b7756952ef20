#!/bin/bash
# round 3, call A: stream-kernel GPU parity tests, then the microbenchmark
OUT=gpurun_out/r3a; mkdir -p $OUT
timeout 600 python -m pytest tests/test_wgrad_stream.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout 600 python scripts/mb_wgrad_stream.py > $OUT/mb.log 2>&1; echo "mb rc=$?" >> $OUT/mb.log
cat $OUT/mb.log | tail -80
