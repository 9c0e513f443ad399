set -x
mkdir -p gpurun_out
timeout -k 10 300 python scripts/bench_matrix.py > gpurun_out/bench_matrix.log 2>&1; cat gpurun_out/bench_matrix.log
timeout -k 10 600 python bench.py --grid 32768 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_32768_wj.json 2> gpurun_out/bench_32768_wj.err; cat gpurun_out/bench_32768_wj.json; tail -3 gpurun_out/bench_32768_wj.err
