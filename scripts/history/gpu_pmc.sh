#!/bin/bash
TAG=${1:-r01pmc}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$c -o pmc -- python $GRAFT_REPO_ROOT/scripts/pmc_kernels.py > $GRAFT_REPO_ROOT/$OUT/$c.log 2>&1
done
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/SQ -o pmc -- python $GRAFT_REPO_ROOT/scripts/pmc_kernels.py > $GRAFT_REPO_ROOT/$OUT/SQ.log 2>&1
cd $GRAFT_REPO_ROOT; find $OUT -name "*.csv" | head -20; tail -3 $OUT/FETCH_SIZE.log
