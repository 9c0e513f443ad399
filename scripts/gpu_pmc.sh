# HBM traffic of the bench's kernels from PMC counters: separate passes for FETCH_SIZE and WRITE_SIZE
# (MI355X_MICROARCH.md "HBM": they do not fit one pass; FETCH_SIZE under-reports 16-byte streams by 2x on gfx950)
set -x
SM=${1:-wjacobi}
TAG=${2:-r01}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}_${SM}_$C -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --smoother $SM > $R/gpurun_out/pmc_${TAG}_${SM}_$C.log 2>&1
done
ls -R $R/gpurun_out/pmc_${TAG}_${SM}_FETCH_SIZE | head
