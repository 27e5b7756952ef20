#!/bin/bash
# round 2, call A: parity (incl. the reference kernel on the GPU), smoke, default bench line (mixed), per-kernel profile of the
# mixed step (serial eager), a 2-rank RCCL run on ONE GPU (private and shared model).
TAG=${1:-r02a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $OUT/smoke.txt
timeout 900 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_mixed -o madnet -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --repeats 1 --no-graph --no-cpu-baseline --no-roofline --no-paths --no-step-surface --wgrad-lanes 0 > $GRAFT_REPO_ROOT/$OUT/prof_mixed.log 2>&1)
timeout 300 python bench.py --gpus 2 --steps 30 --no-paths --no-cpu-baseline > $OUT/bench_2ranks_1gpu.json 2>$OUT/bench_2ranks.err
timeout 300 python bench.py --gpus 2 --shared-model --steps 30 --no-paths --no-cpu-baseline > $OUT/bench_2ranks_shared_1gpu.json 2>$OUT/bench_2ranks_shared.err
cat $OUT/pytest_gpu.txt | tail -5; cat $OUT/smoke.txt; cut -c1-1500 $OUT/bench_default.json; tail -3 $OUT/bench_2ranks.err; cut -c1-400 $OUT/bench_2ranks_1gpu.json; tail -3 $OUT/bench_2ranks_shared.err; cut -c1-400 $OUT/bench_2ranks_shared_1gpu.json
