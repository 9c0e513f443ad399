# round 2 verification: the whole GPU suite, smoke, the default bench line, then the profile passes
set -x
TAG=${1:-r02f}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/${TAG}_tests.log
tail -4 gpurun_out/${TAG}_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench rc=$?"; tail -c 600 gpurun_out/${TAG}_bench.json
timeout -k 10 300 python bench.py --smoother rb --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_rb.json 2> gpurun_out/${TAG}_bench_rb.err
timeout -k 10 300 python bench.py --config 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_bench_cfg2.json 2>/dev/null
timeout -k 10 300 python scripts/bench_config5.py > gpurun_out/${TAG}_bench_cfg5.json 2>/dev/null
timeout -k 10 300 python scripts/bench_python_call.py > gpurun_out/${TAG}_pycall.log 2>&1
timeout -k 10 400 python scripts/bench_lex.py > gpurun_out/${TAG}_bench_lex.log 2>&1
bash scripts/gpu_r02_profiles.sh $TAG > gpurun_out/${TAG}_profiles.log 2>&1
tail -12 gpurun_out/${TAG}_profiles.log
