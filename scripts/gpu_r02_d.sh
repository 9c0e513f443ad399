# round 2, GPU call D: lex pipeline v3 timings + full GPU suite
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 300 python -m pytest tests/test_lexwave.py -x -q -m gpu > gpurun_out/r02d_lex_tests.log 2>&1
echo "lex tests rc=$?" | tee -a gpurun_out/r02d_lex_tests.log
grep -q "lex tests rc=0" gpurun_out/r02d_lex_tests.log || { tail -20 gpurun_out/r02d_lex_tests.log; exit 1; }
timeout -k 10 300 python scripts/bench_lex.py > gpurun_out/r02d_bench_lex.log 2>&1
echo "bench_lex rc=$?"; cat gpurun_out/r02d_bench_lex.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r02d_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r02d_tests.log
tail -6 gpurun_out/r02d_tests.log
