"""A few GS_LEX V(2,2) cycles at 512^2 for a kernel trace: where the small levels' time goes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan
g = int(os.environ.get("LEX_GRID", "512"))
p = Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), 8, nvec=1)
p.set_shifts([0.0]); p.fill(0, _lib.SLOT_F, 0, 1.0); p.fill(0, _lib.SLOT_V, 0, 0.0)
for _ in range(3):
    p.vcycle(2, 2, _lib.GS_LEX, omega=1.0, nu_coarse=2)
p.sync()
t0 = time.perf_counter()
for _ in range(10):
    p.vcycle(2, 2, _lib.GS_LEX, omega=1.0, nu_coarse=2)
p.sync()
print("ms per cycle", (time.perf_counter() - t0) * 100)
p.close()
