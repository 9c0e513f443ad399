set -x
mkdir -p gpurun_out
TAG=${1:-s2e}
timeout -k 10 300 python scripts/prof_rqmg.py > gpurun_out/rqmg_host_$TAG.log 2>&1; cat gpurun_out/rqmg_host_$TAG.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_rqmg -- python3 $GRAFT_REPO_ROOT/scripts/prof_rqmg.py > $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_rqmg.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_rqmg -name "*kernel_stats.csv" | head -1 | xargs -r head -22
