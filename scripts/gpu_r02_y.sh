set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests/test_lexwave.py tests/test_parity_reference.py tests/test_mehrstellen.py -x -q -m gpu > gpurun_out/r02y_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r02y_tests.log
tail -3 gpurun_out/r02y_tests.log
grep -q "tests rc=0" gpurun_out/r02y_tests.log || exit 1
LEX_MODES=1,0 timeout -k 10 400 python scripts/bench_lex.py > gpurun_out/r02y_bench_lex2.log 2>&1
echo "bench_lex rc=$?"; cat gpurun_out/r02y_bench_lex2.log
