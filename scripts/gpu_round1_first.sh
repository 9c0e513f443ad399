set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_1.log 2>&1 ; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_1.log
tail -5 gpurun_out/pytest_gpu_1.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_1.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_1.log; tail -3 gpurun_out/smoke_1.log
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --grid 4096 > gpurun_out/bench_4096_wj.json 2> gpurun_out/bench_4096_wj.err; cat gpurun_out/bench_4096_wj.json; tail -3 gpurun_out/bench_4096_wj.err
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_16384_wj.json 2> gpurun_out/bench_16384_wj.err; cat gpurun_out/bench_16384_wj.json; tail -3 gpurun_out/bench_16384_wj.err
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --smoother rb > gpurun_out/bench_16384_rb.json 2> gpurun_out/bench_16384_rb.err; cat gpurun_out/bench_16384_rb.json; tail -3 gpurun_out/bench_16384_rb.err
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01_baseline -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_r01_baseline.log 2>&1
ls -R $R/gpurun_out/prof_r01_baseline | head -20
