set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_graph.log 2>&1; tail -5 gpurun_out/pytest_gpu_graph.log
for sm in wjacobi rb; do for g in 16384 4096 1024; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --smoother $sm --grid $g > gpurun_out/bench_graph_${g}_$sm.json 2> gpurun_out/bench_graph_${g}_$sm.err
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_graph_${g}_$sm.json"))
print("$sm $g", "MLUPS %.0f  vcycles/s %.1f  ms/step %.3f" % (d["value"], d["vcycles_per_s"], d["ms_per_step"]))
PY
done; done
