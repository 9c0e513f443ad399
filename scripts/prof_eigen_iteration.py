"""The Rayleigh-quotient iteration of BASELINE config 5 alone (drivers.potential_well_eigensolve, method="vcycle"), for
rocprofv3 --kernel-trace --stats: which kernels an eigen-iteration consists of."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import drivers

g = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
drivers.potential_well_eigensolve(g, cycles=2, method="vcycle")
stats, hist = {}, []
drivers.potential_well_eigensolve(g, cycles=20, method="vcycle", stats=stats, history=hist)
print(json.dumps({"grid": g, "ms_per_iteration": stats["loop_seconds"] / 20 * 1e3, "rho": hist[-1]}))
