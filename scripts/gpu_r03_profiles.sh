# round 3 profiles (same set as round 2 plus the 1-D cycle, the device-resident rqmg cycle and the emulated rank share): rocprofv3 kernel stats of the bench command (wjacobi = the default line, rb), PMC passes for HBM
# traffic (FETCH_SIZE, WRITE_SIZE: separate passes) and for the SQ view of every kernel of the cycle (VALU activity,
# wave cycles, waits), kernel stats of BASELINE configs 2 and 5.  Summaries: scripts/collect_profiles_r03.py.
set -x
TAG=${1:-r03}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
for sm in wjacobi rb; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_$sm -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --smoother $sm > $R/gpurun_out/prof_${TAG}_$sm.log 2>&1
  echo "stats $sm rc=$?"
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}_${sm}_$C -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --smoother $sm > $R/gpurun_out/pmc_${TAG}_${sm}_$C.log 2>&1
    echo "pmc $sm $C rc=$?"
  done
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $R/gpurun_out/pmc_${TAG}_${sm}_SQ -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --smoother $sm > $R/gpurun_out/pmc_${TAG}_${sm}_SQ.log 2>&1
  echo "pmc $sm SQ rc=$?"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_cfg2 -- python3 $R/bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_${TAG}_cfg2.log 2>&1
echo "cfg2 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_cfg5 -- python3 $R/scripts/bench_config5.py > $R/gpurun_out/prof_${TAG}_cfg5.log 2>&1
echo "cfg5 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_1d -- python3 $R/bench.py --no-cpu-baseline --skip other,mehrstellen,lex,scaling,configs > $R/gpurun_out/prof_${TAG}_1d.log 2>&1
echo "1d rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_rqmg -- python3 $R/scripts/bench_rqmg.py 8192 2 > $R/gpurun_out/prof_${TAG}_rqmg.log 2>&1
echo "rqmg rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_share -- python3 $R/bench.py --gpus 1 --emulate-rank 3 --of 8 --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_share.log 2>&1
echo "share rc=$?"
cd $R
git rev-parse HEAD > gpurun_out/prof_${TAG}_commit.txt 2>/dev/null || true
