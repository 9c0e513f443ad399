"""The lexicographic Gauss-Seidel V(2,2) cycle at 4096^2 alone, for rocprofv3 --kernel-trace (scripts/trace_by_grid.py:
which launch of which level the cycle's time belongs to).  usage: prof_lex_cycle.py [grid] [cycles]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan

g = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
p = Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), 8, nvec=1)
p.set_shifts([0.0]); p.fill(0, _lib.SLOT_F, 0, 1.0); p.fill(0, _lib.SLOT_V, 0, 0.0)
for _ in range(3):
    p.vcycle(2, 2, _lib.GS_LEX, omega=1.0, nu_coarse=2)
p.sync(); t0 = time.perf_counter()
for _ in range(n):
    p.vcycle(2, 2, _lib.GS_LEX, omega=1.0, nu_coarse=2)
p.sync()
print(json.dumps({"grid": g, "V22_ms": (time.perf_counter() - t0) / n * 1e3, "cycles_run": n + 3}))
p.download(3, _lib.SLOT_V, 0)
p.close()
