"""Does what ran before in the process change the emulated rank share?  (bench.py's default line measured 3.8 ms where a
fresh process measures 2.2.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multigridcmt_amd import _lib, dist_bench
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan

what = sys.argv[1] if len(sys.argv) > 1 else "none"
print("fresh:", dist_bench.time_rank_share(32768, 2, 8, "rb", 3, 8)["ms_per_rank_share"], flush=True)
if what in ("big", "both"):
    big = Plan(laplacian_operator(32768, "2d") * (-1.0 / np.pi ** 2), 8, nvec=1, device=0)
    big.set_shifts([0.0])
    big.fill(0, _lib.SLOT_F, 0, 1.0)
    big.fill(0, _lib.SLOT_V, 0, 0.0)
    for _ in range(4):
        big.vcycle(2, 2, _lib.GS_MC, omega=1.0, k=1, nu_coarse=2)
    big.sync()
    big.close()
    print("after a 32768^2 plan:", dist_bench.time_rank_share(32768, 2, 8, "rb", 3, 8)["ms_per_rank_share"], flush=True)
if what in ("graphs", "both"):
    p = Plan(laplacian_operator(4096, "2d") * (-1.0 / np.pi ** 2), 8, nvec=1, device=0)
    p.set_shifts([0.0])
    p.fill(0, _lib.SLOT_F, 0, 1.0)
    for _ in range(6):
        p.vcycle(2, 2, _lib.GS_LEX, omega=1.0, k=1, nu_coarse=2)
    p.sync()
    p.close()
    print("after a lexicographic plan:", dist_bench.time_rank_share(32768, 2, 8, "rb", 3, 8)["ms_per_rank_share"], flush=True)
print("again:", dist_bench.time_rank_share(32768, 2, 8, "rb", 3, 8)["ms_per_rank_share"], flush=True)
