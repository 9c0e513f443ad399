# round 2, session 2: prefetch-set rotation of the 9-point fused passes (variants/lib_sets9_*.so against the default)
set -x
mkdir -p gpurun_out
TAG=${1:-sets}
timeout -k 10 500 python scripts/tune_levels.py > gpurun_out/levels_$TAG.log 2>&1; cat gpurun_out/levels_$TAG.log
timeout -k 10 500 python scripts/tune_cycles.py > gpurun_out/cycles_$TAG.log 2>&1; cat gpurun_out/cycles_$TAG.log
