"""Rank program of tests/test_distributed.py::test_bench_under_torchrun: bench.py's multi-GPU leg started by the SAME
launcher command the round driver uses (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1 --master-port P ...), on CPU: gloo instead of RCCL, the emulated kernels instead of the GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "hip_cpu_mock")):
    if p not in sys.path:
        sys.path.insert(0, p)
import build_emu                                    # noqa: E402
from multigridcmt_amd import _lib, dist_bench       # noqa: E402
import bench                                        # noqa: E402

_lib.use_library(build_emu.build())
args = bench.parse(sys.argv[1:])
assert int(os.environ["WORLD_SIZE"]) == args.gpus
args.transport = "torch"
dist_bench.run(args, backend="gloo", on_gpu=False)
