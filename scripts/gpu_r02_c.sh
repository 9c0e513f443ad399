# round 2, third GPU call: lex wave pipeline v2 (tests + timings), world-1 RCCL test, seams, staged transfers
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 300 python -m pytest tests/test_lexwave.py -x -q -m gpu > gpurun_out/r02c_lex_tests.log 2>&1
echo "lex tests rc=$?" | tee -a gpurun_out/r02c_lex_tests.log
tail -5 gpurun_out/r02c_lex_tests.log
grep -q "lex tests rc=0" gpurun_out/r02c_lex_tests.log || exit 1
timeout -k 10 300 python scripts/bench_lex.py > gpurun_out/r02c_bench_lex.log 2>&1
echo "bench_lex rc=$?"; cat gpurun_out/r02c_bench_lex.log
timeout -k 10 600 python -m pytest tests/test_full_size_gpu.py tests/test_parity_reference.py tests/test_edge_cases.py -x -q -m gpu -k "sharded or seam or mutated or rqmin" > gpurun_out/r02c_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r02c_tests.log
tail -15 gpurun_out/r02c_tests.log
timeout -k 10 300 python scripts/bench_python_call.py > gpurun_out/r02c_pycall_staged.log 2>&1
echo "pycall rc=$?"; head -4 gpurun_out/r02c_pycall_staged.log
MGCMT_STAGED_TRANSFER=0 timeout -k 10 300 python scripts/bench_python_call.py > gpurun_out/r02c_pycall_plain.log 2>&1
head -4 gpurun_out/r02c_pycall_plain.log
