#!/bin/bash
# The whole multi-GPU job in one call: private-model and shared-model lines at N = 1 / 2 / 4 / 8 (weak scaling: one stream per
# rank), S = 4 private streams per GPU, MAD shared-model.  One JSON line per run -> gpurun_out/scale/*.json (+ scale_summary.txt).
#   usage (on a node with N GPUs): bash scripts/gpu_scale.sh [max_gpus=8] [steps=50] [warmup=10]
# Every line carries rccl.{world,version,ranks[*].{name,arch,cus,xccs,pci_bus_id}}; the shared lines add shared_model.collective_ms_alone,
# collective_ms_in_step (step with - step without the all-reduces) and algbw_gbs.  Scaling efficiency is NOT computed here (the driver does that).
MAXG=${1:-8}; STEPS=${2:-50}; WARM=${3:-10}
OUT=$PWD/gpurun_out/scale; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
HAVE=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
COMMON="--steps $STEPS --warmup $WARM --no-cpu-baseline --no-roofline --no-paths --no-step-surface --drift-steps 0"
PORT=29610
run() {   # tag, n, extra args
  local tag=$1 n=$2; shift 2
  if [ "$n" -gt "$HAVE" ]; then echo "$tag: skipped ($n GPUs wanted, $HAVE present)" | tee -a $OUT/scale_summary.txt; return; fi
  PORT=$((PORT + 1))
  if [ "$n" -eq 1 ]; then
    timeout 900 python bench.py --gpus 1 $COMMON "$@" > $OUT/$tag.json 2> $OUT/$tag.err
  else
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $n $COMMON "$@" > $OUT/$tag.json 2> $OUT/$tag.err
  fi
  python - "$OUT/$tag.json" "$tag" <<'PY' | tee -a $OUT/scale_summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    sm = d.get("shared_model") or {}
    print("%-22s n=%d  %9.1f %s  %.3f ms/step%s" % (sys.argv[2], d["n_gpus"], d["value"], d["unit"], d["ms_per_step"],
          ("  collective alone %.3f ms, in step %.3f ms, %.1f GB/s" % (sm.get("collective_ms_alone", 0), sm.get("collective_ms_in_step", 0), sm.get("algbw_gbs") or 0)) if sm else ""))
except Exception as e:
    print("%-22s FAILED (%s)" % (sys.argv[2], e))
PY
}
: > $OUT/scale_summary.txt
for n in 1 2 4 8; do
  [ "$n" -gt "$MAXG" ] && break
  run private_n$n $n
  run shared_n$n $n --shared-model
done
for n in $(printf '1\n%s\n' "$MAXG" | sort -un); do
  run private_s4_n$n $n --concurrent-streams 4
done
run mad_shared_n$MAXG $MAXG --mode MAD --shared-model
run mad_private_n1 1 --mode MAD
cat $OUT/scale_summary.txt
