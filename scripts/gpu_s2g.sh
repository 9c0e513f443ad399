set -x
mkdir -p gpurun_out
TAG=${1:-s2g}
timeout -k 10 600 python -m pytest tests/test_drivers.py tests/test_parity_reference.py -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -3 gpurun_out/pytest_gpu_$TAG.log
timeout -k 10 300 python scripts/bench_config5.py > gpurun_out/cfg5_$TAG.json 2>&1; cat gpurun_out/cfg5_$TAG.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_eigen -- python3 $GRAFT_REPO_ROOT/scripts/prof_eigen_iteration.py > $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_eigen.log 2>&1
grep ms_per_iteration $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_eigen.log
find $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_eigen -name "*kernel_stats.csv" | head -1 | xargs -r head -12
