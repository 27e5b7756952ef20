#!/bin/bash
TAG=${1:-r02g}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_parity.py -m gpu -q -k level_front 2>&1 | grep -E "^E  |assert|passed|failed" | cut -c1-300 | head -40
