set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
for sm in wjacobi rb; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_big_$sm -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --smoother $sm > $R/gpurun_out/prof_big_$sm.log 2>&1
done
