"""Third bisect: does it matter WHEN librccl enters the process?  argv[1]: none | uid (dlopen + ncclGetUniqueId first)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multigridcmt_amd import _lib, benchdata, dist_bench
from multigridcmt_amd.distributed import rccl_unique_id
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan

mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if mode == "uid":
    rccl_unique_id()
S = -1.0 / np.pi ** 2
g = 16384
plan = Plan(laplacian_operator(g, "2d") * S, 8, nvec=1, device=0)
plan.set_shifts([0.0])
plan.fill(0, _lib.SLOT_F, 0, 1.0)
plan.fill(0, _lib.SLOT_V, 0, 0.0)
for _ in range(5):
    plan.vcycle(2, 2, _lib.WJACOBI, omega=2 / 3, k=1, nu_coarse=2)
plan.sync()
if mode == "keep":
    print("share with the 16384^2 plan alive %.3f" % dist_bench.time_rank_share(32768, 2, 8, "rb", 3, 8)["ms_per_rank_share"], flush=True)
plan.close()
print(mode, "share at the end %.3f" % dist_bench.time_rank_share(32768, 2, 8, "rb", 3, 8)["ms_per_rank_share"], flush=True)
print(mode, "and again %.3f" % dist_bench.time_rank_share(32768, 2, 8, "rb", 3, 8)["ms_per_rank_share"], flush=True)
