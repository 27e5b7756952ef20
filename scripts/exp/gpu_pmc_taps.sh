#!/bin/bash
# SQ / traffic counters of the all-taps filter-gradient kernel next to the tiled one (separate --pmc passes, --kernel-trace only; MI355X_MICROARCH.md rocprofv3 section).
# The per-dispatch rows appear in launch order: per B (1, 4): 5 x tiled, 5 x taps, 5 x taps skeleton.
TAG=${1:-r03taps}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
P="python $GRAFT_REPO_ROOT/scripts/pmc_kernels_taps.py"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$c -o pmc -- $P > $GRAFT_REPO_ROOT/$OUT/$c.log 2>&1
done
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/SQ -o pmc -- $P > $GRAFT_REPO_ROOT/$OUT/SQ.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/SQ2 -o pmc -- $P > $GRAFT_REPO_ROOT/$OUT/SQ2.log 2>&1
timeout 200 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/SQ3 -o pmc -- $P > $GRAFT_REPO_ROOT/$OUT/SQ3.log 2>&1
cd $GRAFT_REPO_ROOT
for d in FETCH_SIZE WRITE_SIZE SQ SQ2 SQ3; do echo "== $d"; tail -2 $OUT/$d.log; f=$(find $OUT/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = r.get("Kernel_Name", "")[:60]
    if "wgrad" not in k: continue
    key = (k, r.get("Counter_Name"))
    agg.setdefault(key, []).append(float(r.get("Counter_Value", 0)))
for (k, c), v in agg.items():
    print("%-62s %-28s n=%2d  first5 %s  last5 %s" % (k, c, len(v), ["%.3g" % x for x in v[:5]], ["%.3g" % x for x in v[-5:]]))
PY
done
