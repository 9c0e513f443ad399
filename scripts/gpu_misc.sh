set -x
mkdir -p gpurun_out
timeout -k 10 300 python scripts/bench_matrix.py > gpurun_out/bench_matrix.log 2>&1; cat gpurun_out/bench_matrix.log
timeout -k 10 600 python bench.py --grid 32768 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_32768_wj.json 2> gpurun_out/bench_32768_wj.err; cat gpurun_out/bench_32768_wj.json; tail -3 gpurun_out/bench_32768_wj.err
timeout -k 10 600 python bench.py --grid 32768 --steps 5 --warmup 1 --no-cpu-baseline --smoother rb > gpurun_out/bench_32768_rb.json 2> gpurun_out/bench_32768_rb.err; cat gpurun_out/bench_32768_rb.json
timeout -k 10 300 python bench.py --grid 4096 --steps 50 --warmup 5 --no-cpu-baseline --smoother rb > gpurun_out/bench_4096_rb.json 2> gpurun_out/bench_4096_rb.err; cat gpurun_out/bench_4096_rb.json
timeout -k 10 400 python scripts/bench_config5.py 8192 > gpurun_out/config5_misc.log 2>&1; cat gpurun_out/config5_misc.log
