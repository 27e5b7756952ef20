#!/bin/bash
# round 6, call y: is the streamed filter gradient bound by its transposing LDS reads?  per-op table with the taps of a row sharing one operand (16 instead of 40 reads per step: 101),
# without the MFMAs (102), both (103)
OUT=gpurun_out/r6y; mkdir -p $OUT
for t in 0 101 102 103; do
  timeout 300 python scripts/plan_table.py --tune wgrad_stream=$t > $OUT/plan_table_$t.txt 2>&1
  echo "tune $t"; grep -n "wgrad_stream_kernel<\|wgrad_stream_mixed" $OUT/plan_table_$t.txt | sed -n 1,3p | cut -c1-110; grep -n "kind 29" $OUT/plan_table_$t.txt | cut -c1-60 | tr '\n' ';'; echo
done
