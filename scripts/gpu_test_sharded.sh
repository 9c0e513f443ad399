set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -k "sharded" > gpurun_out/pytest_sharded.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_sharded.log; tail -30 gpurun_out/pytest_sharded.log
