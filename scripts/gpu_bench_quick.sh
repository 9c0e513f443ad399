set -x
mkdir -p gpurun_out
TAG=${1:-x}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1 ; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -4 gpurun_out/pytest_gpu_$TAG.log
for sm in wjacobi rb; do
  for g in 16384 4096; do
    timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --smoother $sm --grid $g > gpurun_out/bench_${TAG}_${g}_$sm.json 2> gpurun_out/bench_${TAG}_${g}_$sm.err; python - <<PY
import json
d=json.load(open("gpurun_out/bench_${TAG}_${g}_$sm.json"))
print("$sm $g", "MLUPS %.0f  vcycles/s %.1f  ms/step %.3f  smoother_mlups %.0f  roofline %.3f" % (d["value"], d["vcycles_per_s"], d["ms_per_step"], d["smoother_mlups"], d["roofline"]["frac"]))
PY
    tail -2 gpurun_out/bench_${TAG}_${g}_$sm.err
  done
done
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for sm in wjacobi rb; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_$sm -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --smoother $sm > $R/gpurun_out/prof_${TAG}_$sm.log 2>&1
done
