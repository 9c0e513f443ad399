"""Second bisect of bench.py's in-process rank share (3.9 ms against 2.2 fresh): uploads, residual loops, the big plan."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multigridcmt_amd import _lib, benchdata, dist_bench
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan


def share(tag):
    print(tag, "%.3f" % dist_bench.time_rank_share(32768, 2, 8, "rb", 3, 8)["ms_per_rank_share"], flush=True)


S = -1.0 / np.pi ** 2
V, F, T = (_lib.SLOT_V, 0), (_lib.SLOT_F, 0), (_lib.SLOT_T, 0)
g = 16384
f = benchdata.rhs(g)
share("fresh")
plan = Plan(laplacian_operator(g, "2d") * S, 8, nvec=1, device=0)
plan.set_shifts([0.0])
plan.upload(0, _lib.SLOT_F, 0, f)
share("after a 2 GiB upload")
plan.fill(0, _lib.SLOT_V, 0, 0.0)
for _ in range(3):
    plan.vcycle(2, 2, _lib.WJACOBI, omega=2 / 3, k=1, nu_coarse=2)
    plan.apply(0, V, T, with_shift=True)
    plan.axpy(0, -1.0, F, T)
    plan.dot(0, T, T)
share("after a residual loop")
plan.close()
share("after closing the 16384^2 plan")
big = Plan(laplacian_operator(2 * g, "2d") * S, 8, nvec=1, device=0)
big.set_shifts([0.0])
big.upload(1, _lib.SLOT_F, 0, f)
big.prolong(0, F, F)
big.fill(0, _lib.SLOT_V, 0, 0.0)
for _ in range(3):
    big.vcycle(2, 2, _lib.GS_MC, omega=1.0, k=1, nu_coarse=2)
big.sync()
share("with the 32768^2 plan alive")
big.close()
share("after closing it")
