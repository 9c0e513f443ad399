set -x
mkdir -p gpurun_out
TAG=${1:-t2}
timeout -k 10 300 python scripts/tune_options.py > gpurun_out/options_$TAG.log 2>&1; cat gpurun_out/options_$TAG.log
timeout -k 10 400 python scripts/tune_cycles.py > gpurun_out/cycles_$TAG.log 2>&1; cat gpurun_out/cycles_$TAG.log
