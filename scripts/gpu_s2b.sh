# session 2: zero-start flag — GPU suite, config 5 timings, kernel mix of the eigen-iteration
set -x
mkdir -p gpurun_out
TAG=${1:-s2b}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -4 gpurun_out/pytest_gpu_$TAG.log
timeout -k 10 300 python scripts/bench_config5.py > gpurun_out/cfg5_$TAG.json 2>&1; cat gpurun_out/cfg5_$TAG.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_eigen -- python3 $GRAFT_REPO_ROOT/scripts/prof_eigen_iteration.py > $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_eigen.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_eigen.log | tail -3
find $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_eigen -name "*kernel_stats.csv" | head -1 | xargs -r head -25
