"""Rows-per-chunk sweep of the fused passes, level by level (MGCMT_OPT_FUSED_ROWS): one down pass (two sweeps,
residual, restriction; mode 2 + zero start on the Galerkin levels as the cycle issues it) and one up pass (prolong,
two sweeps; mode 1) per level of a 4096 x 4096 plan, for each chunk length.  Prints one JSON line per level."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan

GRID = int(os.environ.get("SWEEP_GRID", "4096"))
KINDS = {"rb": _lib.GS_MC, "wj": _lib.WJACOBI}
ROWS = (0, 4, 6, 8, 12, 16, 24, 32, 48, 64, 128)


def timed(p, fn, n=200):
    for _ in range(5):
        fn()
    p.sync(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    p.sync(); return round((time.perf_counter() - t0) / n * 1e6, 2)


p = Plan(laplacian_operator(GRID, "2d") * (-1 / np.pi ** 2), 8, nvec=1)
p.set_option(_lib.OPT_GRAPH, 0)
p.set_shifts([0.0])
for l in range(p.num_levels):
    p.fill(l, _lib.SLOT_F, 0, 1.0); p.fill(l, _lib.SLOT_V, 0, 0.0)
for name, kind in KINDS.items():
    omega = 1.0 if name == "rb" else 0.8
    for l in range(0, min(6, p.num_levels - 1)):
        if p.fused_max_sweeps(l, kind) < 2:
            continue
        row = {"smoother": name, "level": l, "grid": GRID >> l, "down_us": {}, "up_us": {}}
        for r in ROWS:
            if r > (GRID >> l):
                continue
            p.set_option(_lib.OPT_FUSED_ROWS, r)
            row["down_us"][r] = timed(p, lambda: p.fused_pass(l, kind, 2, omega, 2 + (4 if l else 0)))
            row["up_us"][r] = timed(p, lambda: p.fused_pass(l, kind, 2, omega, 1))
        print(json.dumps(row), flush=True)
p.close()
