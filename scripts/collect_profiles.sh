#!/bin/bash
# copies the summaries of an artifact run (gpurun_out/<tag>, scripts/gpu_final.sh) into profiles/${ROUND}_* (tracked): bash scripts/collect_profiles.sh <tag>
ROUND=${ROUND:-r06}; SRC=gpurun_out/${1:-${ROUND}final}
for f in $SRC/bench_*.json; do b=$(basename $f .json); cp $f profiles/${ROUND}_${b/bench_/bench_line_}.json; done
for f in $SRC/*_graph_kernel_stats.csv $SRC/graph_timeline_*.txt $SRC/plan_table_*.txt $SRC/planes_phases_step.txt $SRC/kernel_resources.txt $SRC/smoke.txt; do
  [ -f $f ] && cp $f profiles/${ROUND}_$(basename $f)
done
[ -f $SRC/microbench_corr.txt ] && cp $SRC/microbench_corr.txt profiles/${ROUND}_microbench_corr.txt
[ -f $SRC/pytest_gpu.txt ] && cp $SRC/pytest_gpu.txt profiles/${ROUND}_pytest_gpu.txt
ls profiles | grep "^${ROUND}_" | wc -l
