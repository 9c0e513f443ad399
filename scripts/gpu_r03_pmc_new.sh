# round 3: HBM traffic (FETCH_SIZE, WRITE_SIZE: separate passes) of the kernels new this round — the fused 1-D passes
# (scripts/bench_1d.py) and the two rqmin passes (scripts/bench_rqmg.py); summary: scripts/collect_pmc_new.py
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for what in 1d rqmg; do
  if [ $what = 1d ]; then CMD="$R/scripts/bench_1d.py 26"; else CMD="$R/scripts/bench_rqmg.py 8192 2"; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_r03_${what}_$C -- python3 $CMD > $R/gpurun_out/pmc_r03_${what}_$C.log 2>&1
    echo "pmc $what $C rc=$?"
  done
done
