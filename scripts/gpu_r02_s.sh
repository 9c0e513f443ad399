set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r02s_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r02s_tests.log
tail -4 gpurun_out/r02s_tests.log
grep -q "tests rc=0" gpurun_out/r02s_tests.log || exit 1
timeout -k 10 400 python scripts/bench_lex.py > gpurun_out/r02s_bench_lex.log 2>&1
echo "bench_lex rc=$?"; cat gpurun_out/r02s_bench_lex.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
