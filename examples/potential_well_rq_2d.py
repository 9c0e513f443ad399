"""BASELINE config 5: ground state of the 2-D square potential well by Rayleigh-quotient minimisation with a V-cycle
preconditioner (and, for comparison, the reference's Rayleigh-quotient multigrid carried to 2-D).  Combines
RQMin.py:15-50 with the potential of PotWellSolver.py:150-153.  usage: potential_well_rq_2d.py [gridsize] [cycles]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigridcmt_amd import drivers  # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 else 2 ** 9
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for method in ("vcycle", "rqmg"):
    hist = []
    start = time.perf_counter()
    rho, _ = drivers.potential_well_eigensolve(g, depth=50.0, cycles=cycles, method=method, nu=2, lowest=8, history=hist)
    elapsed = time.perf_counter() - start
    print("%s  grid %d^2  %d cycles  %.3f s  rho per cycle: %s" % (method, g, cycles, elapsed, " ".join("%.12g" % r for r in hist)))
