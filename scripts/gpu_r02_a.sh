# round 2, first GPU call: the oracle-at-size parity tests, then the default bench line
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu -k "oracle or closed_form" > gpurun_out/r02a_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r02a_tests.log
tail -5 gpurun_out/r02a_tests.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err
echo "bench rc=$?"
tail -c 3000 gpurun_out/r02a_bench.json
