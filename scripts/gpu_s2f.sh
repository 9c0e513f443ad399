set -x
mkdir -p gpurun_out
TAG=${1:-s2f}
timeout -k 10 600 python -m pytest tests/test_fused_kernels.py tests/test_full_size_gpu.py -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -4 gpurun_out/pytest_gpu_$TAG.log
for i in 1 2; do timeout -k 10 300 python scripts/tune_cycles.py >> gpurun_out/cycles_$TAG.log 2>&1; done; cat gpurun_out/cycles_$TAG.log
timeout -k 10 300 python scripts/bench_config5.py > gpurun_out/cfg5_$TAG.json 2>&1; grep "vcycle_ms\|per_iter" gpurun_out/cfg5_$TAG.json
