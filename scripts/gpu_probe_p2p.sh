set -x
mkdir -p gpurun_out
timeout -k 10 200 python scripts/probe_p2p.py > gpurun_out/probe_p2p.log 2>&1; cat gpurun_out/probe_p2p.log
