set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
# -s: a GPU fault's message goes to stderr, which pytest's capture would swallow
timeout -k 10 600 python -m pytest tests/test_lexwave.py tests/test_full_size_gpu.py -x -q -s -m gpu -k "lex or sharded" > gpurun_out/r02j_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r02j_tests.log
tail -3 gpurun_out/r02j_tests.log
grep -q "tests rc=0" gpurun_out/r02j_tests.log || exit 1
timeout -k 10 300 python scripts/bench_lex.py > gpurun_out/r02j_bench_lex.log 2>&1
echo "bench_lex rc=$?"; cat gpurun_out/r02j_bench_lex.log
grep -q speedup gpurun_out/r02j_bench_lex.log || exit 1
timeout -k 10 300 python scripts/sweep_rows.py > gpurun_out/r02j_sweep_rows.log 2>&1
echo "sweep rc=$?"; cat gpurun_out/r02j_sweep_rows.log
timeout -k 10 300 python scripts/bench_python_call.py > gpurun_out/r02j_pycall.log 2>&1
head -3 gpurun_out/r02j_pycall.log
timeout -k 10 300 python bench.py --gpus 1 --force-sharded --steps 10 --warmup 3 --grid 16384 --smoother rb > gpurun_out/r02j_sharded1.json 2> gpurun_out/r02j_sharded1.err
echo "sharded rc=$?"; cut -c1-300 gpurun_out/r02j_sharded1.json
