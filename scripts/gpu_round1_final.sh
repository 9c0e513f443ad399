set -x
mkdir -p gpurun_out
TAG=${1:-r01f}
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1 ; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -4 gpurun_out/pytest_gpu_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; tail -2 gpurun_out/smoke_$TAG.log
timeout -k 10 400 python bench.py > gpurun_out/bench_${TAG}_default.json 2> gpurun_out/bench_${TAG}_default.err; cat gpurun_out/bench_${TAG}_default.json; tail -2 gpurun_out/bench_${TAG}_default.err
timeout -k 10 400 python bench.py --smoother rb --no-cpu-baseline > gpurun_out/bench_${TAG}_rb.json 2> gpurun_out/bench_${TAG}_rb.err; cat gpurun_out/bench_${TAG}_rb.json
bash scripts/gpu_pmc.sh wjacobi $TAG > gpurun_out/pmc_run_wj.log 2>&1
bash scripts/gpu_pmc.sh rb $TAG > gpurun_out/pmc_run_rb.log 2>&1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for sm in wjacobi rb; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_$sm -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --smoother $sm > $R/gpurun_out/prof_${TAG}_$sm.log 2>&1
done
