#!/bin/bash
# Round-end artifact run: serial (lane-less) eager rocprofv3 kernel stats for bf16 and fp32, default bench line, fp32 / MAD /
# DispNet / batched / shared-model bench lines, microbenchmarks.  Outputs under gpurun_out/$TAG (copied to profiles/ by hand).
TAG=${1:-final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_bf16.json
timeout 300 python bench.py --precision fp32 2>/dev/null | tail -1 > $OUT/bench_fp32.json
timeout 300 python bench.py --mode MAD 2>/dev/null | tail -1 > $OUT/bench_mad.json
timeout 300 python bench.py --model dispnet --steps 30 2>/dev/null | tail -1 > $OUT/bench_dispnet_bf16.json
timeout 300 python bench.py --model dispnet --precision fp32 --steps 30 2>/dev/null | tail -1 > $OUT/bench_dispnet_fp32.json
timeout 300 python bench.py --streams-per-gpu 4 --steps 30 2>/dev/null | tail -1 > $OUT/bench_batched4.json
timeout 300 python bench.py --streams-per-gpu 8 --steps 20 2>/dev/null | tail -1 > $OUT/bench_batched8.json
timeout 300 python bench.py --shared-model --steps 30 2>/dev/null | tail -1 > $OUT/bench_shared_model_1gpu.json
for P in bf16 fp32; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$P -o madnet -- python $GRAFT_REPO_ROOT/bench.py --precision $P --steps 10 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-parity-path --wgrad-lanes 0 > $GRAFT_REPO_ROOT/$OUT/prof_$P.log 2>&1)
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_default -o madnet -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity-path > $GRAFT_REPO_ROOT/$OUT/prof_default.log 2>&1)
timeout 300 python scripts/microbench.py corr 2>&1 | grep -v amdgpu.ids > $OUT/microbench_corr.txt
timeout 300 python scripts/microbench.py conv bf16 2>&1 | grep -v amdgpu.ids > $OUT/microbench_conv_bf16.txt
for f in $OUT/bench_*.json; do echo "$f: $(cut -c1-260 $f)"; done
