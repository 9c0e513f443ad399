set -x
mkdir -p gpurun_out
TAG=${1:-c5}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1 ; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -4 gpurun_out/pytest_gpu_$TAG.log
timeout -k 10 400 python scripts/bench_config5.py 8192 > gpurun_out/config5_$TAG.log 2>&1; head -12 gpurun_out/config5_$TAG.log
timeout -k 10 300 python scripts/tune_cycles.py > gpurun_out/cycles_$TAG.log 2>&1; cat gpurun_out/cycles_$TAG.log
