set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests/test_lexwave.py -x -q -s -m gpu > gpurun_out/r02t_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r02t_tests.log
tail -3 gpurun_out/r02t_tests.log
grep -q "tests rc=0" gpurun_out/r02t_tests.log || exit 1
timeout -k 10 400 python scripts/bench_lex.py > gpurun_out/r02t_bench_lex.log 2>&1
echo "bench_lex rc=$?"; cat gpurun_out/r02t_bench_lex.log
