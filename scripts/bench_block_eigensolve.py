"""SURVEY par. 8(f)4: the blocked Rayleigh-Ritz eigen-solver (drivers.block_eigensolve) at size: k = 4 lowest pairs of
-laplacian/pi^2 (exact eigenvalues known) and of the 2-D square well of BASELINE config 5; time per iteration."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import drivers
from multigridcmt_amd.operators import laplacian_operator, potential_well_operator

for g in (1024, 4096, 8192):
    op = laplacian_operator(g, "2d") * (-1 / np.pi ** 2)
    exact = drivers.exact_box_eigenvalues(g, "2d", 4)
    for use_p in (True, False):
        hist, stats = [], {}
        drivers.block_eigensolve(op, k=4, cycles=2, lowest=8)                      # plan, kernels
        vals, _ = drivers.block_eigensolve(op, k=4, cycles=12, lowest=8, history=hist, stats=stats, use_p=use_p)
        print(json.dumps({"operator": "laplacian", "grid": g, "k": 4, "method": "lobpcg" if use_p else "block steepest descent",
                          "ms_per_iteration": round(stats["loop_seconds"] / 12 * 1e3, 3),
                          "errors_after_12": [float("%.2e" % e) for e in np.abs(vals - exact)],
                          "errors_after_6": [float("%.2e" % e) for e in np.abs(hist[5] - exact)]}), flush=True)
g = 8192
op = potential_well_operator(g, 50.0, (g // 4, 3 * g // 4))
hist, stats = [], {}
drivers.block_eigensolve(op, k=4, cycles=2, lowest=8)
vals, _ = drivers.block_eigensolve(op, k=4, cycles=12, lowest=8, history=hist, stats=stats)
print(json.dumps({"operator": "square well (config 5)", "grid": g, "k": 4, "method": "lobpcg",
                  "ms_per_iteration": round(stats["loop_seconds"] / 12 * 1e3, 3), "ritz_values": [float("%.10f" % v) for v in vals],
                  "last_change": [float("%.2e" % e) for e in np.abs(hist[-1] - hist[-2])]}), flush=True)
