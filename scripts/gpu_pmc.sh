#!/bin/bash
# HBM traffic + SQ counters of every conv / filter-gradient / correlation op of the recorded plans (MADNet FULL, MAD blocks, DispNet FULL) and of the fixed roofline
# entries: separate --pmc passes, --kernel-trace only (MI355X_MICROARCH.md HBM / rocprofv3 section) -> profiles/${ROUND}_pmc_roofline.json
#   bash scripts/gpu_pmc.sh <tag>          ROUND=r06 (default) names the output file; PMC_TUNE="conv_bank_small=0,..." applies library tuning hooks first;
#   PMC_PASSES="FETCH_SIZE WRITE_SIZE" skips the SQ pass
export ROUND=${ROUND:-r06}
TAG=${1:-${ROUND}pmc}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export PMC_OPS_JSON=$GRAFT_REPO_ROOT/$OUT/ops.json
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$c -o pmc -- python $GRAFT_REPO_ROOT/scripts/pmc_plan.py > $GRAFT_REPO_ROOT/$OUT/$c.log 2>&1
done
[ "${PMC_PASSES:-all}" = "all" ] && timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/SQ -o pmc -- python $GRAFT_REPO_ROOT/scripts/pmc_plan.py > $GRAFT_REPO_ROOT/$OUT/SQ.log 2>&1
cd $GRAFT_REPO_ROOT; tail -2 $OUT/FETCH_SIZE.log $OUT/SQ.log
python scripts/pmc_summarize.py $OUT | tail -70
cp profiles/${ROUND}_pmc_roofline${PMC_SUFFIX}.json $OUT/
find $OUT -type f -size +3M -delete
