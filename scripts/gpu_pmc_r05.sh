#!/bin/bash
# HBM traffic + SQ counters of every conv / filter-gradient / correlation op of the recorded plans (MADNet FULL, MAD blocks, DispNet FULL) and of the fixed roofline
# entries: separate --pmc passes, --kernel-trace only (MI355X_MICROARCH.md HBM / rocprofv3 section) -> profiles/r05_pmc_roofline.json
TAG=${1:-r05pmc}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export PMC_OPS_JSON=$GRAFT_REPO_ROOT/$OUT/ops.json
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$c -o pmc -- python $GRAFT_REPO_ROOT/scripts/pmc_plan_r05.py > $GRAFT_REPO_ROOT/$OUT/$c.log 2>&1
done
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/SQ -o pmc -- python $GRAFT_REPO_ROOT/scripts/pmc_plan_r05.py > $GRAFT_REPO_ROOT/$OUT/SQ.log 2>&1
cd $GRAFT_REPO_ROOT; tail -2 $OUT/FETCH_SIZE.log $OUT/SQ.log
python scripts/pmc_summarize_r05.py $OUT | tail -70
cp profiles/r05_pmc_roofline.json $OUT/
find $OUT -type f -size +3M -delete
