"""Device time of one cycle of the reference's Rayleigh-quotient multigrid (MGCMTSolver.vcycle_rqmg, MGCMTSolver.py:99-122,
with the 2-D transfers) on the 2-D square well of BASELINE config 5, the iterate resident on the GPU: rqmin as two passes
per step (csrc/kernels_rq.hip).  usage: bench_rqmg.py [grid] [nu] [cycles-only]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
from multigridcmt_amd.operators import identity_operator, potential_well_operator
from multigridcmt_amd.solver import MGCMTSolver

g = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nu = int(sys.argv[2]) if len(sys.argv) > 2 else 4
op, M = potential_well_operator(g, 50.0, (g // 4, 3 * g // 4)), identity_operator(g, "2d")
S = MGCMTSolver()
plan = S._rq_plan(op, M, 8)
plan.set_shifts(np.zeros(S._RQ_REGS))
plan.upload(0, _lib.SLOT_V, S._X, np.random.RandomState(0).random_sample(g * g))
rhos = []
for _ in range(2):
    rhos.append(S._rqmg_levels(plan, 0, nu, nu)[1])
plan.sync()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    rhos.append(S._rqmg_levels(plan, 0, nu, nu)[1])
plan.sync()
ms = (time.perf_counter() - t0) / n * 1e3
if len(sys.argv) > 3:  # (for kernel traces: nothing but cycles)
    print(json.dumps({"ms_per_cycle": ms, "cycles_run": 2 + n}))
    sys.exit(0)
# one rqmin call alone on the finest level: nu steps of 64 B per point + the initial pair (x read twice, g written: 24 B)
plan.sync()
t0 = time.perf_counter()
for _ in range(n):
    S._rqmin_device(plan, 0, nu, want_rho=False)
plan.sync()
ms_fine = (time.perf_counter() - t0) / n * 1e3
pts = float(g) * g
print(json.dumps({"workload": "vcycle_rqmg on the %d^2 square well, nu1 = nu2 = %d, nmin 8, iterate resident" % (g, nu), "ms_per_cycle": ms,
                  "rayleigh_quotients": rhos, "rqmin_fine_level_ms": ms_fine,
                  "rqmin_fine_level_GBs_of_compulsory_bytes": (nu * 64.0 + 24.0) * pts / (ms_fine * 1e-3) / 1e9}))
