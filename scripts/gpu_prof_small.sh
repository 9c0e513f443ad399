set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
for sm in wjacobi rb; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_small_$sm -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --smoother $sm --grid 1024 > $R/gpurun_out/prof_small_$sm.log 2>&1
done
