#!/bin/bash
# HBM traffic + SQ counters of the round-2 roofline kernels: separate --pmc passes, --kernel-trace only (MI355X_MICROARCH.md HBM / rocprofv3 section)
TAG=${1:-r02pmc}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$c -o pmc -- python $GRAFT_REPO_ROOT/scripts/pmc_kernels_r02.py > $GRAFT_REPO_ROOT/$OUT/$c.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/SQ -o pmc -- python $GRAFT_REPO_ROOT/scripts/pmc_kernels_r02.py > $GRAFT_REPO_ROOT/$OUT/SQ.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/SQ2 -o pmc -- python $GRAFT_REPO_ROOT/scripts/pmc_kernels_r02.py > $GRAFT_REPO_ROOT/$OUT/SQ2.log 2>&1
cd $GRAFT_REPO_ROOT; find $OUT -name "*counter_collection.csv" | head; tail -2 $OUT/FETCH_SIZE.log $OUT/SQ.log $OUT/SQ2.log
python scripts/pmc_summarize_r02.py $OUT
