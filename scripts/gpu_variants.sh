set -x
mkdir -p gpurun_out
TAG=${1:-var}
# correctness of every variant library first (fused vs one-launch-per-operation kernels), then timings
for lib in variants/lib_*.so; do
  MGCMT_TEST_LIBRARY=$PWD/$lib timeout -k 10 600 python -m pytest tests/test_fused_kernels.py -m gpu -x -q > gpurun_out/pytest_$(basename $lib .so)_$TAG.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_$(basename $lib .so)_$TAG.log; tail -3 gpurun_out/pytest_$(basename $lib .so)_$TAG.log
done
timeout -k 10 400 python scripts/tune_cycles.py > gpurun_out/cycles_$TAG.log 2>&1; cat gpurun_out/cycles_$TAG.log
timeout -k 10 400 python scripts/tune_levels.py > gpurun_out/levels_$TAG.log 2>&1; cat gpurun_out/levels_$TAG.log
