# round 2, second GPU call: the whole GPU suite (new: RCCL world-1 path, lex wave pipeline), lex timings, the sharded
# bench leg at world 1 through RCCL, kernel tables for configs 2 and 5
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 300 python -m pytest tests/test_lexwave.py -x -q -m gpu > gpurun_out/r02b_lex_tests.log 2>&1
echo "lex tests rc=$?" | tee -a gpurun_out/r02b_lex_tests.log
tail -5 gpurun_out/r02b_lex_tests.log
grep -q "lex tests rc=0" gpurun_out/r02b_lex_tests.log || exit 1
timeout -k 10 300 python scripts/bench_lex.py > gpurun_out/r02b_bench_lex.log 2>&1
echo "bench_lex rc=$?"; cat gpurun_out/r02b_bench_lex.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02b_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r02b_tests.log
tail -15 gpurun_out/r02b_tests.log
timeout -k 10 300 python bench.py --gpus 1 --force-sharded --steps 5 --warmup 2 --grid 16384 --smoother rb > gpurun_out/r02b_sharded1.json 2> gpurun_out/r02b_sharded1.err
echo "sharded rc=$?"; cat gpurun_out/r02b_sharded1.json; tail -5 gpurun_out/r02b_sharded1.err
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_cfg2 -- python3 $R/bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_r02_cfg2.log 2>&1
echo "prof cfg2 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_cfg5 -- python3 $R/scripts/bench_config5.py > $R/gpurun_out/prof_r02_cfg5.log 2>&1
echo "prof cfg5 rc=$?"
ls -R $R/gpurun_out/prof_r02_cfg2 | head
