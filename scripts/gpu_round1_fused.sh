set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_2.log 2>&1 ; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_2.log
tail -15 gpurun_out/pytest_gpu_2.log
for sm in wjacobi rb; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --smoother $sm > gpurun_out/bench2_16384_$sm.json 2> gpurun_out/bench2_16384_$sm.err; cat gpurun_out/bench2_16384_$sm.json; tail -3 gpurun_out/bench2_16384_$sm.err
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --smoother $sm --grid 4096 > gpurun_out/bench2_4096_$sm.json 2> gpurun_out/bench2_4096_$sm.err; cat gpurun_out/bench2_4096_$sm.json
done
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for sm in wjacobi rb; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01_fused_$sm -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --smoother $sm > $R/gpurun_out/prof_r01_fused_$sm.log 2>&1
done
