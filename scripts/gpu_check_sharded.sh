set -x
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -k sharded > gpurun_out/pytest_sharded.log 2>&1; tail -30 gpurun_out/pytest_sharded.log
timeout -k 10 400 python bench.py > gpurun_out/bench_r01g_default.json 2> gpurun_out/bench_r01g_default.err; cat gpurun_out/bench_r01g_default.json; tail -2 gpurun_out/bench_r01g_default.err
nproc; python -c "import os; print(os.cpu_count(), len(os.sched_getaffinity(0)))"
