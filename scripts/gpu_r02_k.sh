set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests/test_full_size_gpu.py tests/test_abi_and_host.py -x -q -s -m gpu -k "transfers or recycle or abi" > gpurun_out/r02k_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r02k_tests.log
tail -3 gpurun_out/r02k_tests.log
grep -q "tests rc=0" gpurun_out/r02k_tests.log || exit 1
timeout -k 10 300 python scripts/bench_python_call.py > gpurun_out/r02k_pycall.log 2>&1
echo "pycall rc=$?"; head -3 gpurun_out/r02k_pycall.log
