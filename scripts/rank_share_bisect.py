"""Which leg of bench.py's default line changes the emulated rank share measured after it?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multigridcmt_amd import _lib, dist_bench
from multigridcmt_amd.operators import laplacian_operator, mehrstellen_operator
from multigridcmt_amd.plan import Plan


def share(tag):
    print(tag, "%.3f" % dist_bench.time_rank_share(32768, 2, 8, "rb", 3, 8)["ms_per_rank_share"], flush=True)


S = -1.0 / np.pi ** 2
share("fresh")
g = 16384
plan = Plan(laplacian_operator(g, "2d") * S, 8, nvec=1, device=0)
plan.set_shifts([0.0])
plan.fill(0, _lib.SLOT_F, 0, 1.0)
plan.fill(0, _lib.SLOT_V, 0, 0.0)
for _ in range(12):
    plan.vcycle(2, 2, _lib.WJACOBI, omega=2 / 3, k=1, nu_coarse=2)
plan.sync()
share("after headline cycles")
plan.time_smoother(0, _lib.WJACOBI, 2, 2 / 3, 25)
plan.time_fused_pass(0, _lib.WJACOBI, 2, 2 / 3, 10, 20)
plan.time_fused_pass(0, _lib.WJACOBI, 2, 2 / 3, 33, 20)
share("after pass timings")
for k_ in (0, 1, 2):
    plan.bandwidth_probe(0, k_, 1024, 5)
share("after bandwidth probes")
plan.close()
m9 = Plan(mehrstellen_operator(g) * S, 8, nvec=1, device=0)
m9.set_shifts([0.0])
m9.fill(0, _lib.SLOT_F, 0, 1.0)
for _ in range(5):
    m9.vcycle(2, 2, _lib.GS_MC, omega=1.0, k=1, nu_coarse=2)
m9.sync()
m9.close()
share("after mehrstellen")
lx = Plan(laplacian_operator(4096, "2d") * S, 8, nvec=1, device=0)
lx.set_shifts([0.0])
lx.fill(0, _lib.SLOT_F, 0, 1.0)
for _ in range(5):
    lx.vcycle(2, 2, _lib.GS_LEX, omega=1.0, k=1, nu_coarse=2)
lx.set_option(_lib.OPT_LEX_WAVE, 0)
lx.vcycle(2, 2, _lib.GS_LEX, omega=1.0, k=1, nu_coarse=2)
lx.sync()
lx.close()
share("after lexicographic")
p1 = Plan(laplacian_operator(1 << 24, "1d") * S, 8, nvec=1, device=0)
p1.set_shifts([0.0])
p1.fill(0, _lib.SLOT_F, 0, 1.0)
for _ in range(5):
    p1.vcycle(2, 2, _lib.WJACOBI, omega=2 / 3, k=1, nu_coarse=2)
p1.set_option(_lib.OPT_FUSED, 0)
for _ in range(3):
    p1.vcycle(2, 2, _lib.WJACOBI, omega=2 / 3, k=1, nu_coarse=2)
p1.sync()
p1.close()
share("after 1-D")
