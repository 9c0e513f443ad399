# session 2, last verification: whole GPU suite, smoke, the default bench line, the drop-in call with host arrays, the examples
set -x
mkdir -p gpurun_out
TAG=${1:-s2h}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -3 gpurun_out/pytest_gpu_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
timeout -k 10 300 python scripts/bench_python_call.py > gpurun_out/${TAG}_pycall.log 2>&1; head -3 gpurun_out/${TAG}_pycall.log
bash scripts/gpu_examples.sh > gpurun_out/${TAG}_examples.log 2>&1; grep "rc=" gpurun_out/${TAG}_examples.log
