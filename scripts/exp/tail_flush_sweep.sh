mkdir -p gpurun_out/r4n; B="--no-paths --no-cpu-baseline --no-roofline --no-step-surface --drift-steps 0 --repeats 3 --steps 100 --stamps 20"
run() { n=$1; shift; python bench.py $B "$@" 2>gpurun_out/r4n/$n.err | tail -1 > gpurun_out/r4n/$n.json; }
run base
run prog432 --set "engine.PYR_FLUSH_BEFORE=(4,3,2)"
run b4_b2 --set "engine.PYR_FLUSH_AFTER=(9,1)" --set "engine.PYR_FLUSH_BEFORE=(4,2)"
run b4_b3 --set "engine.PYR_FLUSH_AFTER=(9,1)" --set "engine.PYR_FLUSH_BEFORE=(4,3)"
run b5_b3 --set "engine.PYR_FLUSH_AFTER=(9,1)" --set "engine.PYR_FLUSH_BEFORE=(5,3)"
run a5_b3 --set "engine.PYR_FLUSH_BEFORE=(3,)"
run a5_b2 --set "engine.PYR_FLUSH_BEFORE=(2,)"
run a5_b4 --set "engine.PYR_FLUSH_BEFORE=(4,)"
run base2
