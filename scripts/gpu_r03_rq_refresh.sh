# round 3, after the rqmin rework: kernel stats + per-level table of one vcycle_rqmg cycle, the square-well eigen-iteration's
# timeline, and the PMC passes of the rq kernels (summaries: profiles/r03_rqmg_*, r03_eigen_iteration_*, r03_pmc_new_kernels.json)
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_rqmg -- python3 $R/scripts/bench_rqmg.py 8192 2 > $R/gpurun_out/prof_r03_rqmg.log 2>&1
echo "rqmg stats rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r03_rqmg_cycles -o t -- python3 $R/scripts/bench_rqmg.py 8192 2 cycles > $R/gpurun_out/prof_r03_rqmg_cycles.log 2>&1
echo "rqmg cycles rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r03_eigen -o t -- python3 $R/scripts/prof_eigen_iteration.py 8192 > $R/gpurun_out/prof_r03_eigen.log 2>&1
echo "eigen rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_r03_rqmg_$C -- python3 $R/scripts/bench_rqmg.py 8192 2 > $R/gpurun_out/pmc_r03_rqmg_$C.log 2>&1
  echo "pmc rqmg $C rc=$?"
done
cd $R
python3 scripts/bench_rqmg.py 8192 2 > gpurun_out/r03_rqmg_plain.json
python3 scripts/bench_rqmg.py 8192 4 > gpurun_out/r03_rqmg_plain_nu4.json
python3 scripts/prof_eigen_iteration.py 8192 > gpurun_out/r03_eigen_plain.json
tail -n 2 gpurun_out/r03_rqmg_plain.json gpurun_out/r03_rqmg_plain_nu4.json gpurun_out/r03_eigen_plain.json
