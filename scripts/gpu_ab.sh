set -x
mkdir -p gpurun_out
TAG=${1:-ab}
for i in 1 2 3; do timeout -k 10 300 python scripts/tune_cycles.py >> gpurun_out/cycles_$TAG.log 2>&1; done; cat gpurun_out/cycles_$TAG.log
