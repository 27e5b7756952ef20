#!/bin/bash
# round 3, call B: per-kernel durations of the stream microbenchmark (rocprofv3 kernel trace), B=1 only
OUT=gpurun_out/r3b; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
MB_ONLY_B1=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt -o mb -- python $GRAFT_REPO_ROOT/scripts/mb_wgrad_stream.py > $GRAFT_REPO_ROOT/$OUT/mb.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r3b/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    k = (r["Kernel_Name"][:70], r.get("Grid_Size", r.get("Grid_Size_X","")), r.get("Workgroup_Size", r.get("Workgroup_Size_X","")), r.get("LDS_Block_Size", ""))
    agg.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    v2 = sorted(v)
    print("%-72s grid %-8s wg %-5s lds %-7s n=%3d  med %7.1f us  min %7.1f" % (k[0], k[1], k[2], k[3], len(v), v2[len(v2)//2], v2[0]))
PY
