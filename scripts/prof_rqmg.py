"""Where a cycle of the reference's Rayleigh-quotient multigrid (MGCMTSolver.vcycle_rqmg, RQMin.py:25-27 carried to the
2-D square well of BASELINE config 5) spends its time on the host side: cProfile of two cycles at 8192^2."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd.operators import identity_operator, potential_well_operator
from multigridcmt_amd.solver import MGCMTSolver

g = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
op, M = potential_well_operator(g, 50.0, (g // 4, 3 * g // 4)), identity_operator(g, "2d")
solver = MGCMTSolver()
x = np.random.RandomState(0).random_sample(g * g)
x, rho = solver.vcycle_rqmg(x, op, M, nu1=2, nu2=2, nmin=8)
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter()
for _ in range(2):
    x, rho = solver.vcycle_rqmg(x, op, M, nu1=2, nu2=2, nmin=8)
print("ms per cycle:", (time.perf_counter() - t0) * 500, "rho", rho)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:6000])
