#!/bin/bash
TAG=${1:-r01clk}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/GRBM -o pmc -- python $GRAFT_REPO_ROOT/scripts/pmc_kernels.py > $GRAFT_REPO_ROOT/$OUT/grbm.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_cli_gpu.py tests/test_api_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5
