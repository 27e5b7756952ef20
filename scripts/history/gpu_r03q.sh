#!/bin/bash
# round 2, call AQ: HIP runtime knobs that touch graph execution / signals / kernel arguments (environment only)
TAG=${1:-r03q}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-paths --no-step-surface --no-roofline --steps 100 --repeats 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA 2>/dev/null | tail -1 > $OUT/bench_$name.json; }
run base MH_X=1
run sysscope0 ROC_SYSTEM_SCOPE_SIGNAL=0
run pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run pktcap1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run gq1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run gq2 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run gq4 DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run batch1 DEBUG_HIP_GRAPH_BATCH_SIZE=1
run batch256 DEBUG_HIP_GRAPH_BATCH_SIZE=256
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run fgs1 ROC_USE_FGS_KERNARG=1
run fgs0 ROC_USE_FGS_KERNARG=0
run cpwait1 GPU_STREAMOPS_CP_WAIT=1
run cpwait0 GPU_STREAMOPS_CP_WAIT=0
run dynq0 DEBUG_HIP_DYNAMIC_QUEUES=0
run base2 MH_X=1
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        e=json.load(open(f)); print(f.split("/")[-1], ["%.3f"%x for x in e["timing"]["ms_per_step_all"]])
    except Exception as ex: print(f, "ERR", ex)
PY
