"""V-cycle times with the LDS-resident tail on / off (GPU box)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
from multigridcmt_amd import plan as planmod
from multigridcmt_amd.operators import laplacian_operator
for g in (256, 1024, 4096, 16384):
    p = planmod.Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), 8, nvec=1)
    p.set_shifts([0.0]); p.fill(0, _lib.SLOT_F, 0, 1.0); p.fill(0, _lib.SLOT_V, 0, 0.0)
    row = {"grid": g}
    for tail in (1, 0):
        p.set_option(_lib.OPT_TAIL, tail)
        for name, kind, om in (("wj", _lib.WJACOBI, 2 / 3), ("rb", _lib.GS_MC, 1.0)):
            for _ in range(4): p.vcycle(2, 2, kind, omega=om, nu_coarse=2)
            p.sync(); t0 = time.perf_counter()
            n = 40 if g <= 4096 else 12
            for _ in range(n): p.vcycle(2, 2, kind, omega=om, nu_coarse=2)
            p.sync(); row["%s_tail%d_ms" % (name, tail)] = round((time.perf_counter() - t0) / n * 1e3, 4)
    p.close()
    print(json.dumps(row), flush=True)
