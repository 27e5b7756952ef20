#!/bin/bash
# round 2, late: all-taps filter-gradient kernel -- parity on the MI355X + microbenchmark against the tiled kernel
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_conv_parity.py -m gpu -k "wgrad_taps" -x -q 2>&1 | tail -8 > gpurun_out/r04a_parity.log
timeout 100 python scripts/microbench.py wgradt > gpurun_out/r04a_wgradt.txt 2>&1
cat gpurun_out/r04a_parity.log; cat gpurun_out/r04a_wgradt.txt
